// Developer harness (not part of libaot_hip.so): checks aot_conv2d_nhwc_f32 against a naive fp64-accumulating conv on
// the conv / linear shapes of one R50-AOTL 480p frame (batch 1 and 3) and times every kernel configuration.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/dev/gemm_check.hip \
//         aot-benchmark_amd/csrc/gemm_conv.hip aot-benchmark_amd/csrc/gemm_lds.hip -o tools/dev/gemm_check
//   tools/dev/gemm_check [quick]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/aot_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void naive_conv(const float* in, const float* w, const float* bias, const float* res, float* out, int B, int H,
                           int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dil, int ldb,
                           int res_rows, int act) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long M = (long)B * OH * OW;
  if (idx >= M * Cout) return;
  const int n = idx % Cout;
  const long m = idx / Cout;
  const int b = m / (OH * OW), pix = m % (OH * OW), oy = pix / OW, ox = pix % OW;
  double s = 0.0;
  for (int ky = 0; ky < KH; ++ky)
    for (int kx = 0; kx < KW; ++kx) {
      const int iy = oy * stride - pad + ky * dil, ix = ox * stride - pad + kx * dil;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const float* a = in + (((long)b * H + iy) * W + ix) * Cin;
      const float* ww = w + (long)(ky * KW + kx) * Cin * ldb + n;
      for (int c = 0; c < Cin; ++c) s += (double)a[c] * ww[(long)c * ldb];
    }
  float v = (float)s + (bias ? bias[n] : 0.f);
  if (res) v += res[(res_rows ? m % res_rows : m) * Cout + n];
  if (act == 1) v = fmaxf(v, 0.f);
  out[m * Cout + n] = v;
}

struct Shape { const char* name; int H, W, Cin, Cout, K, s, cnt; };

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  // `gemm_check list 7:117:1,22:117:1` : only the listed (shape index : cfg : batch) launches, five times each, no check --
  // the dispatches a `rocprofv3 --pmc` pass should see
  const char* only = (argc > 2 && !strcmp(argv[1], "list")) ? argv[2] : nullptr;
  const Shape shapes[] = {
      {"l1.c1 64>64", 121, 213, 64, 64, 1, 1, 1},       {"l1.c1 256>64", 121, 213, 256, 64, 1, 1, 2},
      {"l1.c2 3x3 64", 121, 213, 64, 64, 3, 1, 3},      {"l1.c3 64>256", 121, 213, 64, 256, 1, 1, 4},
      {"l2.c1 256>128@4x", 121, 213, 256, 128, 1, 1, 1}, {"l2.c2 3x3s2 128", 121, 213, 128, 128, 3, 2, 1},
      {"l2.c1 512>128", 61, 107, 512, 128, 1, 1, 3},    {"l2.c2 3x3 128", 61, 107, 128, 128, 3, 1, 3},
      {"l2.c3 128>512", 61, 107, 128, 512, 1, 1, 4},    {"l2.ds 256>512 s2", 121, 213, 256, 512, 1, 2, 1},
      {"l3.c1 512>256@8x", 61, 107, 512, 256, 1, 1, 1}, {"l3.c2 3x3s2 256", 61, 107, 256, 256, 3, 2, 1},
      {"l3.c1 1024>256", 31, 54, 1024, 256, 1, 1, 8},   {"l3.c2 3x3 256", 31, 54, 256, 256, 3, 1, 6},
      {"l3.c3 256>1024", 31, 54, 256, 1024, 1, 1, 9},   {"l3.ds 512>1024 s2", 61, 107, 512, 1024, 1, 2, 1},
      {"lstt 256>512", 31, 54, 256, 512, 1, 1, 3},      {"lstt 256>256", 31, 54, 256, 256, 1, 1, 12},
      {"lstt 512>256", 31, 54, 512, 256, 1, 1, 3},      {"dec ad8 512>256", 61, 107, 512, 256, 1, 1, 1},
      {"dec c8 3x3 256>128", 61, 107, 256, 128, 3, 1, 1}, {"dec ad4 256>128", 121, 213, 256, 128, 1, 1, 1},
      {"dec c4 3x3 128", 121, 213, 128, 128, 3, 1, 1},  {"ragged 3x3 d2", 17, 19, 32, 96, 3, 1, 0}};
  // auto | 64x64 register-staged | LDS-direct 64x64 (2 / 3 DMA steps ahead) | lean 64x64 | lean 128x64 | lean 64x64 with
  // split-K 2 / 4 | lean 128x64 with split-K 2 / 4 | wave-independent kernels (in-block split-K 4 / 8, 64x32 waves x4)
  const int cfgs[] = {-1, 4, 117, 133, 197, 213, 198, 200, 214, 216, 14, 18, 24};
  const long scratch_floats = 48L << 20;
  float* scratch;
  CK(hipMalloc(&scratch, scratch_floats * 4));
  if (only) {
    for (const char* q = only; *q;) {
      int si, c, B;
      if (sscanf(q, "%d:%d:%d", &si, &c, &B) != 3) { printf("bad list entry %s\n", q); return 1; }
      const Shape& sh = shapes[si];
      const int pad = sh.K / 2;
      const int OH = (sh.H + 2 * pad - (sh.K - 1) - 1) / sh.s + 1, OW = (sh.W + 2 * pad - (sh.K - 1) - 1) / sh.s + 1;
      const long M = (long)B * OH * OW;
      const int KK = sh.K * sh.K * sh.Cin;
      std::vector<float> hin((size_t)B * sh.H * sh.W * sh.Cin), hw((size_t)KK * sh.Cout);
      unsigned seed = 777u + si;
      auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((int)(seed >> 9) - (1 << 22)) * (1.f / (1 << 22)); };
      for (auto& v : hin) v = rnd();
      for (auto& v : hw) v = rnd() * 0.1f;
      float *din, *dw, *dwt, *dout;
      CK(hipMalloc(&din, hin.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dwt, hw.size() * 4));
      CK(hipMalloc(&dout, (size_t)M * sh.Cout * 4));
      CK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dwt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));     // (values do not matter here)
      int rc = 0;
      CK(hipMemset(scratch, 0, 64));
      for (int i = 0; i < 5; ++i)
        rc = aot_conv2d_nhwc_f32(din, dw, dwt, nullptr, nullptr, dout, scratch, scratch_floats, B, sh.H, sh.W, sh.Cin, OH, OW,
                                 sh.Cout, sh.K, sh.K, sh.s, pad, 1, sh.Cin, sh.Cout, KK, sh.Cout, 0, 0, 1, c, 0);
      CK(hipDeviceSynchronize());
      printf("%s cfg %d batch %d: rc %d, M %ld K %d N %d\n", sh.name, c, B, rc, M, KK, sh.Cout);
#ifdef AOT_LEAN_TIMING
      {   // cycle split of the lean kernel's steps (wave 0 of every workgroup, summed over the five launches)
        unsigned long long t[4];
        CK(hipMemcpy(t, scratch, 32, hipMemcpyDeviceToHost));
        if (t[3]) printf("   per step (s_memtime cycles, %llu steps): DMA/LDS wait %.0f, barrier %.0f, body %.0f\n", t[3],
                         (double)t[0] / t[3], (double)t[1] / t[3], (double)t[2] / t[3]);
      }
#endif
      hipFree(din); hipFree(dw); hipFree(dwt); hipFree(dout);
      while (*q && *q != ',') ++q;
      if (*q == ',') ++q;
    }
    return 0;
  }
  double tot_us[16][2] = {}, tot_gf[2] = {};
  for (int B = 1; B <= 3; B += 2) {
    printf("==== batch %d ====\n%-20s %7s %5s %5s %8s |", B, "shape", "M", "K", "N", "GF");
    for (int c : cfgs) printf(" cfg%4d us   TF |", c);
    printf("\n");
    for (const Shape& sh : shapes) {
      if (quick && sh.cnt == 0 && B == 3) continue;
      const int dil = sh.cnt == 0 ? 2 : 1;
      const int pad = sh.K / 2 * dil;
      const int OH = (sh.H + 2 * pad - dil * (sh.K - 1) - 1) / sh.s + 1, OW = (sh.W + 2 * pad - dil * (sh.K - 1) - 1) / sh.s + 1;
      const long M = (long)B * OH * OW;
      const int KK = sh.K * sh.K * sh.Cin, ldb = sh.Cout;
      std::vector<float> hin((size_t)B * sh.H * sh.W * sh.Cin), hw((size_t)KK * ldb), hwt((size_t)sh.Cout * KK), hb(sh.Cout);
      unsigned seed = 12345u + sh.H * 7 + sh.Cin + sh.Cout * 3 + B;
      auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((int)(seed >> 9) - (1 << 22)) * (1.f / (1 << 22)); };
      for (auto& v : hin) v = rnd();
      for (auto& v : hw) v = rnd() * 0.1f;
      for (auto& v : hb) v = rnd();
      for (int k = 0; k < KK; ++k)
        for (int n = 0; n < sh.Cout; ++n) hwt[(size_t)n * KK + k] = hw[(size_t)k * ldb + n];
      const long res_rows = (long)OH * OW;    // residual map shared by the B images (row = m % res_rows)
      std::vector<float> hres((size_t)res_rows * sh.Cout);
      for (auto& v : hres) v = rnd();
      float *din, *dw, *dwt, *db, *dres, *dout, *dref;
      CK(hipMalloc(&din, hin.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dwt, hwt.size() * 4));
      CK(hipMalloc(&db, hb.size() * 4)); CK(hipMalloc(&dres, hres.size() * 4));
      CK(hipMalloc(&dout, (size_t)M * sh.Cout * 4)); CK(hipMalloc(&dref, (size_t)M * sh.Cout * 4));
      CK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dwt, hwt.data(), hwt.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dres, hres.data(), hres.size() * 4, hipMemcpyHostToDevice));
      const long tot = M * sh.Cout;
      hipLaunchKernelGGL(naive_conv, dim3((tot + 255) / 256), dim3(256), 0, 0, din, dw, db, dres, dref, B, sh.H, sh.W, sh.Cin,
                         OH, OW, sh.Cout, sh.K, sh.K, sh.s, pad, dil, ldb, (int)res_rows, 1);
      CK(hipDeviceSynchronize());
      std::vector<float> href((size_t)tot), hout((size_t)tot);
      CK(hipMemcpy(href.data(), dref, tot * 4, hipMemcpyDeviceToHost));
      const double gf = 2.0 * M * KK * sh.Cout / 1e9;
      printf("%-20s %7ld %5d %5d %8.3f |", sh.name, M, KK, sh.Cout, gf);
      int ci = 0;
      for (int c : cfgs) {
        CK(hipMemset(dout, 0xFF, (size_t)tot * 4));
        auto run = [&]() {
          return aot_conv2d_nhwc_f32(din, dw, dwt, db, dres, dout, scratch, scratch_floats, B, sh.H, sh.W, sh.Cin, OH, OW,
                                     sh.Cout, sh.K, sh.K, sh.s, pad, dil, sh.Cin, ldb, KK, sh.Cout, sh.Cout, (int)res_rows, 1, c, 0);
        };
        const int rc = run();
        if (rc != 0) { printf("       -      - |"); ++ci; continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hout.data(), dout, tot * 4, hipMemcpyDeviceToHost));
        double err = 0;
        for (long i = 0; i < tot; ++i) { const double d = fabs((double)hout[i] - href[i]); if (!(d <= err)) err = d; }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = quick ? 10 : 30;
        for (int i = 0; i < 3; ++i) run();
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) run();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf(" %7.1f %5.1f%s|", us, gf * 1e3 / us, err < 2e-4 ? " " : "!");
        if (!(err < 2e-4)) printf("[ERR %.2e]", err);
        tot_us[ci][B == 3] += us * sh.cnt;
        ++ci;
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
      }
      tot_gf[B == 3] += gf * sh.cnt;
      printf("  x%d\n", sh.cnt);
      hipFree(din); hipFree(dw); hipFree(dwt); hipFree(db); hipFree(dres); hipFree(dout); hipFree(dref);
    }
    printf("per-frame totals (batch %d, %.1f GF):", B, tot_gf[B == 3]);
    for (size_t i = 0; i < sizeof(cfgs) / sizeof(int); ++i)
      printf("  cfg%d %.0f us (%.1f TF)", cfgs[i], tot_us[i][B == 3], tot_gf[B == 3] * 1e3 / tot_us[i][B == 3]);
    printf("\n");
  }
  return 0;
}
