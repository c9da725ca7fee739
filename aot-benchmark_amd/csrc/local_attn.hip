// Short-term (windowed) attention of AOT, one fused kernel.
//
// For query p = (y, x) of head hd, over the (2R+1)^2 window slots w = (dy, dx):
//   s_w  = (q/sqrt(d)) . k[p + delta(w)]  +  ( relk_w[hd, w, :] . q  + relk_b[hd, w] )     (rel on UNSCALED q)
//   a    = softmax_w(s)  over in-image slots (the reference pushes the others to -1e8 -> exactly 0)
//   out  = sum_w a_w * ( v[p + delta(w)] + relv[hd, :, w] )
// This is what the reference computes with the CUDA correlation sampler (or a 386 MB unfold), a 164 MB
// scatter to a dense N x N map and a dense matmul (attention.py:308-428); here nothing is materialised.
//
// Mapping.  A workgroup = 64 consecutive queries of one image row and one head (lane = query) x NWV waves
// that split the 2R+1 window rows round-robin (the map has only h*H ~ 250 such strips, so without the split a
// launch would put one latency-bound wave on each CU).  Each wave walks its window rows with an online
// softmax and the NWV partial (m, l, o) meet in LDS at the end (fixed order -> deterministic).
// Per window row a wave stages the K row (64+2R positions x 32 channels, token-major, 36-float stride ->
// conflict-free ds_read_b128 of each lane's sliding 15-key window) into its private LDS slab, computes its 15
// scores, then overwrites the slab with the V row.  Staging is software pipelined through registers: the V
// row's global loads are issued before the score block, the next K row's before the aggregation block, so L2
// latency hides under VALU work and no workgroup barrier exists in the main loop.
// The relative-position tables are wave-uniform; they are re-laid out as [hd][dy][c][16] so that one VGPR
// (lane%16 = dx) holds a row and DPP row_newbcast feeds 15 FMAs from it.
// The work is ~1.5 GFLOP/layer of irregular 32-long dot products: VALU by nature (fp32 MFMA has the same
// peak and would waste 3/4 of a dense tile).
#include "common.h"


struct LocalParams {
  const float* q;
  const float* k;
  const float* v;
  const float* relk_t;  // [H][WS][32][16]  relk_t[hd][dy][c][dx] = sqrt(d) * relative_emb_k.weight[hd*W2 + dy*WS + dx][c]
  const float* relk_b;  // [H][WS][16]
  const float* relv_t;  // [H][WS][32][16]  relv_t[hd][dy][c][dx] = relative_emb_v[hd][c][dy*WS + dx]
  float* out;
  int h, w, H, ldq, ldk, ldv, ldo;
  float scale_div;
  long kv_brows;   // rows between the k/v maps of consecutive lanes (>= h*w)
};

// Wave-uniform table rows are broadcast through DPP: a table row of 16 floats sits in ONE VGPR with
// lane L holding element L%16 (every 16-lane DPP row has the full copy), and
//     v_fmac_f32_dpp acc, tbl, x row_newbcast:N        acc += tbl[lane N of my row] * x
// reads element N for all lanes at no extra instruction.  (The scalar-cache route costs a ~500-cycle round trip
// per 64 bytes because the 480 KB of tables do not stay in the scalar cache; LDS broadcast reads would double the
// LDS traffic.)  hipcc does not fold v_mov_dpp into the FMA, hence inline asm.  `s_nop 1` opens each statement:
// a VALU write of a DPP source needs 2 wait states and hipcc pads nothing inside asm (guide section 5.7).
#define DPPF(acc, N) "v_fmac_f32_dpp " acc ", %[t], %[x] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
// s[0..14] += tbl[0..14] * x
__device__ __forceinline__ void dpp_axpy15(float (&s)[15], float t, float x) {
  asm("s_nop 1\n\t"
      DPPF("%0", 0) DPPF("%1", 1) DPPF("%2", 2) DPPF("%3", 3) DPPF("%4", 4) DPPF("%5", 5) DPPF("%6", 6) DPPF("%7", 7)
      DPPF("%8", 8) DPPF("%9", 9) DPPF("%10", 10) DPPF("%11", 11) DPPF("%12", 12) DPPF("%13", 13) DPPF("%14", 14)
      : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]), "+v"(s[8]),
        "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14])
      : [t] "v"(t), [x] "v"(x));
}
#undef DPPF
// a0 += sum_dx tbl0[dx] * p[dx],  a1 += sum_dx tbl1[dx] * p[dx]   (two chains interleaved)
#define DPP2(N) "v_fmac_f32_dpp %0, %[t0], %[p" #N "] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                "v_fmac_f32_dpp %1, %[t1], %[p" #N "] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void dpp_dot15x2(float& a0, float& a1, float t0, float t1, const float (&p)[15]) {
  asm("s_nop 1\n\t"
      DPP2(0) DPP2(1) DPP2(2) DPP2(3) DPP2(4) DPP2(5) DPP2(6) DPP2(7) DPP2(8) DPP2(9) DPP2(10) DPP2(11) DPP2(12) DPP2(13) DPP2(14)
      : "+v"(a0), "+v"(a1)
      : [t0] "v"(t0), [t1] "v"(t1), [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]),
        [p5] "v"(p[5]), [p6] "v"(p[6]), [p7] "v"(p[7]), [p8] "v"(p[8]), [p9] "v"(p[9]), [p10] "v"(p[10]),
        [p11] "v"(p[11]), [p12] "v"(p[12]), [p13] "v"(p[13]), [p14] "v"(p[14]));
}
#undef DPP2

template <int R, int NWV>
__global__ void __launch_bounds__(NWV * 64) local_attn_d32_kernel(const LocalParams pin) {
  constexpr int WS = 2 * R + 1, D = 32;
  constexpr int NPOS = 64 + 2 * R;             // staged key positions per row
  constexpr int LDS_LD = 36;
  constexpr int NF4 = NPOS * (D / 4);          // float4 per staged row
  constexpr int PER = (NF4 + 63) / 64;         // float4 per lane
  constexpr int SLAB = (NPOS + 2) * LDS_LD;    // floats per wave slab (>= 64*34 needed by the merge)
  __shared__ __attribute__((aligned(16))) float lds[NWV * SLAB];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // lane b of the batch (object group / clip): its own h x w maps, q/out rows b*N.., k/v rows b*kv_brows..
  const int bl = blockIdx.z / pin.H, hd = blockIdx.z - bl * pin.H;
  LocalParams p = pin;
  {
    const long N = (long)p.h * p.w;
    p.q += bl * N * p.ldq;
    p.out += bl * N * p.ldo;
    p.k += bl * pin.kv_brows * p.ldk;
    p.v += bl * pin.kv_brows * p.ldv;
  }
  const int x0 = blockIdx.x * 64, y = blockIdx.y;
  const bool active = x0 + lane < p.w;
  const int x = active ? x0 + lane : p.w - 1;
  const int n = y * p.w + x;
  float* slab = lds + wave * SLAB;

  // q is kept only in its scaled form q/sqrt(d) (the reference divides, attention.py:330); the relative-position
  // key table arrives pre-multiplied by sqrt(d) (pack_local_tables) so that rel = relk . q is evaluated on the
  // unscaled q as the reference does (attention.py:327), up to one rounding of the table entries.
  float qs[D];
  {
    const float4* src = reinterpret_cast<const float4*>(p.q + (long)n * p.ldq + hd * D);
#pragma unroll
    for (int i = 0; i < D / 4; ++i) {
      const float4 t = src[i];
      qs[4 * i] = t.x / p.scale_div; qs[4 * i + 1] = t.y / p.scale_div;
      qs[4 * i + 2] = t.z / p.scale_div; qs[4 * i + 3] = t.w / p.scale_div;
    }
  }

  float m = -INFINITY, l = 0.f;
  float o[D];
#pragma unroll
  for (int c = 0; c < D; ++c) o[c] = 0.f;

  // window rows of this wave that fall inside the image
  auto row_ok = [&](int dy) { const int ky = y + dy - R; return dy < WS && ky >= 0 && ky < p.h; };
  auto next_row = [&](int dy) { while (dy < WS && !row_ok(dy)) dy += NWV; return dy; };

  float4 stage[PER];
  auto gload = [&](const float* base, int ld, int ky) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = lane + i * 64;
      const int pos = f >> 3, c4 = f & 7;
      const int kx = x0 - R + pos;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < NF4 && kx >= 0 && kx < p.w)
        t = *reinterpret_cast<const float4*>(base + ((long)ky * p.w + kx) * ld + hd * D + c4 * 4);
      stage[i] = t;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = lane + i * 64;
      if (f < NF4) *reinterpret_cast<float4*>(&slab[(f >> 3) * LDS_LD + (f & 7) * 4]) = stage[i];
    }
  };
  // the slab is private to the wave: wave-level ordering of its LDS traffic is all that is needed
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  int dy = next_row(wave);
  if (dy < WS) gload(p.k, p.ldk, y + dy - R);
  while (dy < WS) {
    const int ky = y + dy - R;
    wave_sync();          // previous row's V reads are done
    lstore();             // K row -> LDS
    gload(p.v, p.ldv, ky);  // V row in flight under the score block
    float tk[D];            // rel-pos key table rows of this window row: in flight under the q.k block
    {
      const float* wk = p.relk_t + (((long)hd * WS + dy) * D) * 16 + (lane & 15);
#pragma unroll
      for (int c = 0; c < D; ++c) tk[c] = __builtin_nontemporal_load(wk + c * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    wave_sync();

    // ---- 15 scores: q.k from LDS + relative-position key bias through the scalar cache ----
    float s[WS];
    {
      static_assert(WS == 15, "DPP helpers are written for a 15-wide window");
      const float* bk = p.relk_b + ((long)hd * WS + dy) * 16;
#pragma unroll
      for (int dx = 0; dx < WS; ++dx) s[dx] = bk[dx];
#pragma unroll
      for (int c = 0; c < D; ++c) dpp_axpy15(s, tk[c], qs[c]);
    }
#pragma unroll
    for (int dx = 0; dx < WS; ++dx) {
      float dot = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < D / 4; ++c4) {
        const float4 kk = *reinterpret_cast<const float4*>(&slab[(lane + dx) * LDS_LD + c4 * 4]);
        dot = fmaf(qs[4 * c4], kk.x, dot);
        dot = fmaf(qs[4 * c4 + 1], kk.y, dot);
        dot = fmaf(qs[4 * c4 + 2], kk.z, dot);
        dot = fmaf(qs[4 * c4 + 3], kk.w, dot);
      }
      const int kx = x + dx - R;
      s[dx] = (kx >= 0 && kx < p.w) ? dot + s[dx] : -INFINITY;
    }
    float mt = s[0];
#pragma unroll
    for (int dx = 1; dx < WS; ++dx) mt = fmaxf(mt, s[dx]);
    const float mnew = fmaxf(m, mt);  // finite: the centre column is always inside the image
    const float alpha = expf(m - mnew);
    l *= alpha;
#pragma unroll
    for (int c = 0; c < D; ++c) o[c] *= alpha;
    m = mnew;
#pragma unroll
    for (int dx = 0; dx < WS; ++dx) {
      s[dx] = expf(s[dx] - mnew);  // exp(-inf) = 0 for masked slots
      l += s[dx];
    }

    wave_sync();          // K reads done
    lstore();             // V row -> LDS
    const int dyn = next_row(dy + NWV);
    if (dyn < WS) gload(p.k, p.ldk, y + dyn - R);   // next K row in flight under the aggregation block
    float tv[D];            // rel-pos value table rows: in flight under the p.v block
    {
      const float* rv = p.relv_t + (((long)hd * WS + dy) * D) * 16 + (lane & 15);
#pragma unroll
      for (int c = 0; c < D; ++c) tv[c] = __builtin_nontemporal_load(rv + c * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    wave_sync();

    // ---- aggregation: sum_dx p * v  (LDS)  +  sum_dx p * relv  (scalar cache) ----
#pragma unroll
    for (int dx = 0; dx < WS; ++dx) {
      const float pw = s[dx];
#pragma unroll
      for (int c4 = 0; c4 < D / 4; ++c4) {
        const float4 vv = *reinterpret_cast<const float4*>(&slab[(lane + dx) * LDS_LD + c4 * 4]);
        o[4 * c4] = fmaf(pw, vv.x, o[4 * c4]);
        o[4 * c4 + 1] = fmaf(pw, vv.y, o[4 * c4 + 1]);
        o[4 * c4 + 2] = fmaf(pw, vv.z, o[4 * c4 + 2]);
        o[4 * c4 + 3] = fmaf(pw, vv.w, o[4 * c4 + 3]);
      }
    }
    {
#pragma unroll
      for (int c = 0; c < D; c += 2) dpp_dot15x2(o[c], o[c + 1], tv[c], tv[c + 1], s);
    }
    dy = dyn;
  }

  // ---- merge the NWV partial softmaxes (fixed order) ----
  __syncthreads();
  {
    float* mine = lds + wave * SLAB;     // [34][64]: m, l, o[0..31]
    mine[0 * 64 + lane] = m;
    mine[1 * 64 + lane] = l;
#pragma unroll
    for (int c = 0; c < D; ++c) mine[(2 + c) * 64 + lane] = o[c];
  }
  __syncthreads();
  if (wave == 0 && active) {
    float mm = -INFINITY;
#pragma unroll
    for (int w2 = 0; w2 < NWV; ++w2) mm = fmaxf(mm, lds[w2 * SLAB + lane]);
    float lt = 0.f;
    float ot[D];
#pragma unroll
    for (int c = 0; c < D; ++c) ot[c] = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NWV; ++w2) {
      const float* src = lds + w2 * SLAB;
      const float mw = src[lane];
      const float f = (mw == -INFINITY) ? 0.f : expf(mw - mm);
      lt = fmaf(f, src[64 + lane], lt);
#pragma unroll
      for (int c = 0; c < D; ++c) ot[c] = fmaf(f, src[(2 + c) * 64 + lane], ot[c]);
    }
    const float inv = 1.f / lt;
    float4* dst = reinterpret_cast<float4*>(p.out + (long)n * p.ldo + hd * D);
#pragma unroll
    for (int c4 = 0; c4 < D / 4; ++c4)
      dst[c4] = make_float4(ot[4 * c4] * inv, ot[4 * c4 + 1] * inv, ot[4 * c4 + 2] * inv, ot[4 * c4 + 3] * inv);
  }
}


extern "C" int aot_local_attn_f32(const float* q, const float* k, const float* v, const float* relk_t,
                                  const float* relk_b, const float* relv_t, float* out, int B, long kv_brows, int h,
                                  int w, int H, int d, int max_dis, int ldq, int ldk, int ldv, int ldo, float scale_div,
                                  void* stream) {
  if (!q || !k || !v || !relk_t || !relk_b || !relv_t || !out || h <= 0 || w <= 0 || H <= 0 || B <= 0) return AOT_ERR_BADARG;
  if (B > 1 && kv_brows < (long)h * w) return AOT_ERR_BADARG;
  if ((long)B * H > 65535) return AOT_ERR_UNSUPPORTED;
  if (d != 32 || max_dis != 7) return AOT_ERR_UNSUPPORTED;
  if ((ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3)) return AOT_ERR_BADARG;
  LocalParams p;
  p.q = q; p.k = k; p.v = v; p.relk_t = relk_t; p.relk_b = relk_b; p.relv_t = relv_t; p.out = out;
  p.h = h; p.w = w; p.H = H; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.scale_div = scale_div;
  p.kv_brows = kv_brows;
  hipLaunchKernelGGL((local_attn_d32_kernel<7, 8>), dim3(cdiv(w, 64), h, B * H), dim3(8 * 64), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}
