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

// ---- dependent MFMAs that are NOT back to back (call 3's ISA bisect: s_nop 7 between all MFMAs of the failing kernel made it worse) ----
// MFMA ; W wait states (s_nop, or W independent v_mov as filler) ; MFMA with SrcC = vDst of the first ; ... CH2 times; all sixteen
// accumulator registers are checked per 16-lane quarter.  expected CH2 * 16.
template <int FILL, int W, int CH2>
__global__ void __launch_bounds__(1024) probe_dep(float* out, int iters) {
  float bad[4] = {0.f, 0.f, 0.f, 0.f};
  const int wave = threadIdx.x >> 6;
  const float* src = out + (((long)blockIdx.x * blockDim.x + threadIdx.x) & 0xffff) * 8;      // FILL 2: loads in flight beside the chain
  for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(3);
  for (int it = 0; it < iters; ++it) {
    float r[16];
    asm volatile(ZERO_ACC SET_A SET_B SETTLE
                 ".rept %[ch]\n" MFMA
                 ".if %[fill] == 0\n" ".if %[w] > 0\ns_nop %[k]\n.endif\n" ".else\n" ".rept %[w]\nv_mov_b32 v90, v91\n.endr\n" ".endif\n"
                 ".if %[fill] == 2\nglobal_load_dwordx4 v[92:95], %[ptr], off\nglobal_load_dwordx4 v[96:99], %[ptr], off offset:1024\n.endif\n"
                 ".endr\n" "s_waitcnt vmcnt(0)\n" SETTLE SETTLE
                 "v_mov_b32 %0, v64\nv_mov_b32 %1, v65\nv_mov_b32 %2, v66\nv_mov_b32 %3, v67\nv_mov_b32 %4, v68\nv_mov_b32 %5, v69\n"
                 "v_mov_b32 %6, v70\nv_mov_b32 %7, v71\nv_mov_b32 %8, v72\nv_mov_b32 %9, v73\nv_mov_b32 %10, v74\nv_mov_b32 %11, v75\n"
                 "v_mov_b32 %12, v76\nv_mov_b32 %13, v77\nv_mov_b32 %14, v78\nv_mov_b32 %15, v79\n"
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]),
                   "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15])
                 : [w] "n"(W), [k] "n"(Gap<W>::K), [ch] "n"(CH2), [fill] "n"(FILL), [ptr] "v"(src)
                 : CLOB, "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "memory");
    for (int i = 0; i < 16; ++i)
      if (r[i] != 16.f * CH2) bad[i & 3] += 1.f;
  }
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  out[2 * t] = bad[0] + bad[2];      // even registers
  out[2 * t + 1] = bad[1] + bad[3];  // odd registers
}

template <int FILL, int W, int CH2>
static void run_dep(float* dev, std::vector<float>& host) {
  const int iters = 1000;
  printf("DEP  XDL write -> XDL SrcC same vDst, %s gap %2d wait states, %d in a chain :", FILL == 2 ? "v_mov+loads" : FILL ? "v_mov" : "s_nop", W, CH2);
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps, blocks = 256;
    hipLaunchKernelGGL((probe_dep<FILL, W, CH2>), dim3(blocks), dim3(threads), 0, nullptr, dev, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(host.data(), dev, (size_t)blocks * threads * 2 * 4, hipMemcpyDeviceToHost));
    double q[4] = {0, 0, 0, 0}, ev = 0, od = 0;
    for (long t = 0; t < (long)blocks * threads; ++t) {
      q[(t & 63) >> 4] += host[2 * t] + host[2 * t + 1];
      ev += host[2 * t];
      od += host[2 * t + 1];
    }
    printf("   %d/SIMD: bad regs [q0 %.0f q1 %.0f q2 %.0f q3 %.0f | even %.0f odd %.0f]", wps, q[0], q[1], q[2], q[3], ev, od);
  }
  printf("\n");
}

// ---- the instruction round 4's ISA bisect pinned: an in-place packed add whose op_sel crosses the halves of the overwritten source ----
//     v_pk_add_f32 v[64:65], v[66:67], v[64:65] op_sel:[0,1] op_sel_hi:[1,0]        lo = v66 + v65(old),  hi = v67 + v64(old)
// Half of the waves of a workgroup run it in a loop on lane-dependent values; the other half (BG = 1) keep the SIMD busy the way the
// attention kernel's main loop does (MFMA chain, v_exp_f32, v_perm_b32, packed multiplies) or (BG = 0) stay idle.  Also the
// out-of-place twin (destination = a third pair) and the same-pair form without crossing, as controls.
template <int FORM, int BG, int W, int LD>
__global__ void __launch_bounds__(1024) probe_pkx(float* out, int iters) {
  __shared__ float lds_buf[1024 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  lds_buf[threadIdx.x] = 1.f; lds_buf[threadIdx.x + 1024] = 2.f; lds_buf[threadIdx.x + 2048] = 3.f; lds_buf[threadIdx.x + 3072] = 4.f;
  __syncthreads();
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_buf + threadIdx.x * 4;
  const float* gsrc = out + (((long)blockIdx.x * blockDim.x + threadIdx.x) & 0xffff) * 8 + (1 << 20);
  float badlo = 0.f, badhi = 0.f;
  if ((wave & 1) && BG) {
    for (int it = 0; it < iters; ++it)
      asm volatile(SET_A SET_B ZERO_ACC "v_mov_b32 v90, 0x3f000000\n"
                   ".rept 6\n" MFMA "v_exp_f32 v91, v90\nv_perm_b32 v92, v91, v90, v90\nv_pk_mul_f32 v[94:95], v[92:93], v[90:91]\n.endr\n"
                   : : : CLOB, "v90", "v91", "v92", "v93", "v94", "v95");
  } else if (!(wave & 1) || !BG) {
    for (int it = 0; it < iters; ++it) {
      const float a0 = (float)(lane + 1), a1 = (float)(2 * lane + 3), b0 = (float)(1000 + lane), b1 = (float)(5000 - lane);
      float r0, r1;
      if constexpr (FORM == 0)        // in place, crossed (the unsafe one)
        asm volatile(".if %[ld] == 1\nds_read2st64_b32 v[70:71], %[la] offset1:4\nds_read2st64_b32 v[72:73], %[la] offset0:8 offset1:12\n.endif\n"
                     ".if %[ld] == 2\nglobal_load_dwordx4 v[70:73], %[ga], off\nglobal_load_dwordx4 v[74:77], %[ga], off offset:1024\n.endif\n"
                     "v_mov_b32 v66, %4\nv_mov_b32 v67, %5\nv_mov_b32 v64, %2\nv_mov_b32 v65, %3\n" ".if %[w] > 0\ns_nop %[k]\n.endif\n"
                     "v_pk_add_f32 v[64:65], v[66:67], v[64:65] op_sel:[0,1] op_sel_hi:[1,0]\ns_waitcnt vmcnt(0) lgkmcnt(0)\ns_nop 7\nv_mov_b32 %0, v64\nv_mov_b32 %1, v65\n"
                     : "=v"(r0), "=v"(r1) : "v"(b0), "v"(b1), "v"(a0), "v"(a1), [w] "n"(W), [k] "n"(Gap<W>::K), [ld] "n"(LD), [la] "v"(laddr), [ga] "v"(gsrc)
                     : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "memory");
      else if constexpr (FORM == 1)   // crossed, out of place
        asm volatile(".if %[ld] == 1\nds_read2st64_b32 v[70:71], %[la] offset1:4\nds_read2st64_b32 v[72:73], %[la] offset0:8 offset1:12\n.endif\n"
                     ".if %[ld] == 2\nglobal_load_dwordx4 v[70:73], %[ga], off\nglobal_load_dwordx4 v[74:77], %[ga], off offset:1024\n.endif\n"
                     "v_mov_b32 v66, %4\nv_mov_b32 v67, %5\nv_mov_b32 v64, %2\nv_mov_b32 v65, %3\n" ".if %[w] > 0\ns_nop %[k]\n.endif\n"
                     "v_pk_add_f32 v[68:69], v[66:67], v[64:65] op_sel:[0,1] op_sel_hi:[1,0]\ns_waitcnt vmcnt(0) lgkmcnt(0)\ns_nop 7\nv_mov_b32 %0, v68\nv_mov_b32 %1, v69\n"
                     : "=v"(r0), "=v"(r1) : "v"(b0), "v"(b1), "v"(a0), "v"(a1), [w] "n"(W), [k] "n"(Gap<W>::K), [ld] "n"(LD), [la] "v"(laddr), [ga] "v"(gsrc)
                     : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "memory");
      else                            // in place, not crossed: lo = v66 + v64, hi = v67 + v65
        asm volatile(".if %[ld] == 1\nds_read2st64_b32 v[70:71], %[la] offset1:4\nds_read2st64_b32 v[72:73], %[la] offset0:8 offset1:12\n.endif\n"
                     ".if %[ld] == 2\nglobal_load_dwordx4 v[70:73], %[ga], off\nglobal_load_dwordx4 v[74:77], %[ga], off offset:1024\n.endif\n"
                     "v_mov_b32 v66, %4\nv_mov_b32 v67, %5\nv_mov_b32 v64, %2\nv_mov_b32 v65, %3\n" ".if %[w] > 0\ns_nop %[k]\n.endif\n"
                     "v_pk_add_f32 v[64:65], v[66:67], v[64:65]\ns_waitcnt vmcnt(0) lgkmcnt(0)\ns_nop 7\nv_mov_b32 %0, v64\nv_mov_b32 %1, v65\n"
                     : "=v"(r0), "=v"(r1) : "v"(b0), "v"(b1), "v"(a0), "v"(a1), [w] "n"(W), [k] "n"(Gap<W>::K), [ld] "n"(LD), [la] "v"(laddr), [ga] "v"(gsrc)
                     : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "memory");
      const float w0 = FORM == 2 ? a0 + b0 : a0 + b1, w1 = FORM == 2 ? a1 + b1 : a1 + b0;
      if (r0 != w0) badlo += 1.f;
      if (r1 != w1) badhi += 1.f;
    }
  }
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  out[2 * t] = badlo;
  out[2 * t + 1] = badhi;
}

template <int FORM, int BG, int W, int LD>
static void run_pkx(float* dev, std::vector<float>& host, const char* what) {
  const int iters = 4000;
  printf("PKX  %s, %d wait states behind its producers, %s, other waves %s :", what, W, LD == 1 ? "2 LDS reads landing" : LD == 2 ? "2 global loads landing" : "no loads", BG ? "busy" : "idle");
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps, blocks = 512;
    hipLaunchKernelGGL((probe_pkx<FORM, BG, W, LD>), dim3(blocks), dim3(threads), 0, nullptr, dev, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(host.data(), dev, (size_t)blocks * threads * 2 * 4, hipMemcpyDeviceToHost));
    double lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    for (long t = 0; t < (long)blocks * threads; ++t) { lo[(t & 63) >> 4] += host[2 * t]; hi[(t & 63) >> 4] += host[2 * t + 1]; }
    printf("   %d/SIMD: wrong low halves by lane quarter [%.0f %.0f %.0f %.0f] high [%.0f %.0f %.0f %.0f]", wps, lo[0], lo[1], lo[2], lo[3], hi[0], hi[1],
           hi[2], hi[3]);
  }
  printf("\n");
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* dev;
  CK(hipMalloc(&dev, (size_t)512 * 1024 * 2 * 4 + (1 << 23)));
  std::vector<float> host((size_t)512 * 1024 * 2);
  if (argc > 1 && argv[1][0] == 'p') {       // `mfma_hazard_probe pkx`
#define PKX(W, LD) run_pkx<0, 0, W, LD>(dev, host, "in place, crossed    "); run_pkx<0, 1, W, LD>(dev, host, "in place, crossed    "); \
  run_pkx<1, 1, W, LD>(dev, host, "out of place, crossed"); run_pkx<2, 1, W, LD>(dev, host, "in place, straight   ");
    PKX(0, 1) PKX(1, 1) PKX(2, 1) PKX(4, 1) PKX(8, 1) PKX(12, 1) PKX(16, 1) PKX(24, 1) PKX(0, 2) PKX(2, 2) PKX(8, 2) PKX(0, 0)
    return 0;
  }
#define DEP(W) run_dep<0, W, 4>(dev, host); run_dep<1, W, 4>(dev, host); run_dep<2, W, 4>(dev, host);
  DEP(0) DEP(1) DEP(2) DEP(3) DEP(4) DEP(5) DEP(6) DEP(7) DEP(8) DEP(9) DEP(10) DEP(11) DEP(12) DEP(14) DEP(16)
  if (argc > 1) return 0;       // `mfma_hazard_probe dep`: only the dependent-chain rows
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
