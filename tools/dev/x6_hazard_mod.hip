// Runs hand-edited builds of one bf16x6 attention kernel (code objects made by tools/dev/isa_bisect.py) against the shipped kernel:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/dev/x6_hazard_mod.hip -o tools/dev/x6_hazard_mod
//   tools/dev/x6_hazard_mod tools/dev/bisect/KERNEL tools/dev/bisect/e*.co
// For every code object: REPS launches on the same inputs, each compared bit for bit with the product kernel's result; for the first
// differing launch the wrong entries are localised: which (key split, query tile, head) tiles, which query rows of the tile, which
// channels, a sample of values.
#include "../../aot-benchmark_amd/csrc/attention_x6.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  if (argc < 3) { printf("usage: x6_hazard_mod KERNEL_NAME_FILE a.co [b.co ...]\n"); return 1; }
  char kname[512] = {0};
  { FILE* f = fopen(argv[1], "r"); if (!f || !fgets(kname, sizeof kname, f)) { printf("cannot read %s\n", argv[1]); return 1; } fclose(f); kname[strcspn(kname, "\r\n")] = 0; }
  const int reps = 20;
  const int N = 1674, H = 8, C = 256, M = 4, ns = 3;
  const int T = M * N - 13;
  const long cap = ((long)M * N + 31) / 32 * 32;
  std::vector<float> hq((size_t)N * C), hk((size_t)M * N * C), hv((size_t)M * N * C);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
  for (auto& x : hq) x = 3.f * rnd();
  for (auto& x : hk) x = 2.f * rnd();
  for (auto& x : hv) x = 2.f * rnd();
  float *q, *k, *v, *out, *ref, *part, *pref;
  unsigned short* kv;
  const size_t kvbytes = (size_t)(cap / 32) * H * 6144 * 2;
  const size_t partn = (size_t)ns * N * (C + 2 * H);
  CK(hipMalloc(&q, hq.size() * 4)); CK(hipMalloc(&k, hk.size() * 4)); CK(hipMalloc(&v, hv.size() * 4));
  CK(hipMalloc(&out, (size_t)N * C * 4)); CK(hipMalloc(&ref, (size_t)N * C * 4)); CK(hipMalloc(&kv, kvbytes));
  CK(hipMalloc(&part, partn * 4)); CK(hipMalloc(&pref, partn * 4));
  CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(k, hk.data(), hk.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(v, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(kv, 0, kvbytes));
  for (int slot = 0; slot < M; ++slot)
    if (aot_attn_pack_x6_f32(k + (size_t)slot * N * C, v + (size_t)slot * N * C, kv, 1, N, C, 0, C, C, cap, nullptr, slot, nullptr)) return 1;
  AttnX6Params p;
  p.q = q; p.kv = kv; p.out = ref; p.part = pref; p.T_dev = nullptr; p.Nq = N; p.T = T; p.H = H; p.ldq = C; p.ldo = C;
  p.nsplit = ns; p.B = 1; p.cap_rows = cap; p.scale_div = 5.656854249492381f;
  CK(hipMemset(pref, 0, partn * 4));
  if (aot_attn_x6_f32(q, kv, ref, pref, 1, cap, N, T, nullptr, H, 32, C, C, p.scale_div, ns, nullptr)) return 1;
  CK(hipDeviceSynchronize());
  std::vector<float> h0(partn), h1(partn);
  CK(hipMemcpy(h0.data(), pref, partn * 4, hipMemcpyDeviceToHost));
  p.out = out; p.part = part;
  const size_t on = (size_t)ns * N * C;
  const int ntile = (T + 31) / 32, tps = (ntile + ns - 1) / ns, tpw = (tps + 3) / 4;
  printf("kernel %s\nbank of %d frames (T = %d, %d key tiles), key split %d: %d tiles per workgroup, %d per wave; %d launches per build\n", kname, M, T,
         ntile, ns, tps, tpw, reps);
  for (int a = 2; a < argc; ++a) {
    hipModule_t mod;
    hipFunction_t fn;
    if (hipModuleLoad(&mod, argv[a]) != hipSuccess || hipModuleGetFunction(&fn, mod, kname) != hipSuccess) { printf("%s: cannot load\n", argv[a]); continue; }
    int bad_launches = 0;
    size_t bad_vals = 0, bad_ml = 0;
    bool shown = false;
    std::string detail;
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(part, 0, partn * 4, nullptr));
      void* args[] = {&p};
      CK(hipModuleLaunchKernel(fn, H, (N + 31) / 32, ns, 256, 1, 1, 0, nullptr, args, nullptr));
      CK(hipMemcpy(h1.data(), part, partn * 4, hipMemcpyDeviceToHost));
      size_t nb = 0;
      std::map<long, std::set<int>> tiles;      // (split, qt, head) -> query rows j of the tile
      std::map<long, std::set<int>> chans;      // ... -> channels of the head
      char sample[256] = {0};
      for (size_t i = 0; i < partn; ++i)
        if (memcmp(&h0[i], &h1[i], 4)) {
          ++nb;
          if (i >= on) { ++bad_ml; continue; }
          const int c = (int)(i % C), row = (int)((i / C) % N), sp = (int)(i / ((size_t)C * N));
          const long key = ((long)sp * 64 + row / 32) * 8 + c / 32;
          tiles[key].insert(row & 31);
          chans[key].insert(c & 31);
          if (!sample[0]) snprintf(sample, sizeof sample, "first: split %d row %d ch %d: got %.6g want %.6g", sp, row, c, h1[i], h0[i]);
        }
      if (nb) ++bad_launches;
      bad_vals += nb;
      if (nb && !shown) {
        shown = true;
        char buf[4096];
        int o = snprintf(buf, sizeof buf, "      launch %d: %zu values in %zu (split, query tile, head) tiles; %s\n", r, nb, tiles.size(), sample);
        int shown_t = 0;
        for (auto& kvp : tiles) {
          if (shown_t++ >= 6) break;
          const long key = kvp.first;
          const int head = (int)(key & 7), qt = (int)((key >> 3) & 63), sp = (int)(key >> 9);
          o += snprintf(buf + o, sizeof buf - o, "        split %d query tile %2d head %d: query rows j = {", sp, qt, head);
          for (int j : kvp.second) o += snprintf(buf + o, sizeof buf - o, "%d ", j);
          o += snprintf(buf + o, sizeof buf - o, "}  channels {");
          for (int c : chans[key]) o += snprintf(buf + o, sizeof buf - o, "%d ", c);
          o += snprintf(buf + o, sizeof buf - o, "}\n");
        }
        detail = buf;
      }
    }
    printf("  %-44s launches differing %2d / %d   values %7zu  (of them m / l entries: %zu)\n", strrchr(argv[a], '/') ? strrchr(argv[a], '/') + 1 : argv[a],
           bad_launches, reps, bad_vals, bad_ml);
    if (shown) printf("%s", detail.c_str());
    CK(hipModuleUnload(mod));
  }
  return 0;
}
