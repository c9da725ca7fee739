// Short-term gated propagation of DeAOT (LocalGatedPropagation.forward, reference attention.py:789-861):
// one head, q = k of width 128, a value of width C = 1024 ([V | ID_V]) and a gate u:
//   s_w  = (q/sqrt(128)) . k[p + delta(w)] + relk[w, :] . q + relk_b[w]       over the 15x15 window
//   a    = softmax_w(s)   (window slots outside the image excluded)
//   out  = ( sum_w a_w * v[p + delta(w)] ) * u
// The value is 32x wider than the key, so the work is split in three launches that share the slot-major
// probability map P [225, N] (1.5 MB at 480p, L2 resident):
//   lgp_scores_kernel    window scores -> P (raw, -inf outside the image)
//   lgp_softmax_kernel   in-place softmax over the 225 slots of every query
//   lgp_aggregate_kernel sum_w P * v over 32-channel chunks of the value, times the gate
// Staging, LDS layout (token-major rows, 36-float stride, conflict-free ds_read_b128), the split of the window
// rows over the waves of a workgroup and the DPP row_newbcast broadcast of the wave-uniform relative-position
// table are those of local_attn.hip.
#include "common.h"

struct LgpParams {
  const float* q;       // [N, ldq] (128 wide)
  const float* k;       // [N, ldk]
  const float* v;       // [N, ldv] (C wide)
  const float* gate;    // [N, ldg] or null
  const float* relk_t;  // [WS][128][16]  sqrt(128) * relative_emb_k.weight[dy*WS+dx][c]
  const float* relk_b;  // [WS][16]
  float* prob;          // [WS*WS][N]
  float* out;           // [N, ldo]
  int h, w, C, ldq, ldk, ldv, ldg, ldo;
  float scale_div;
  long kv_brows;        // rows between the k/v maps of consecutive lanes (>= h*w)
};

// lane b of the batch (object group / clip): its own maps and its own [225, N] probability scratch
__device__ __forceinline__ LgpParams lgp_lane(const LgpParams& pin, int bl) {
  LgpParams p = pin;
  const long N = (long)p.h * p.w;
  p.q += bl * N * p.ldq;
  p.k += bl * pin.kv_brows * p.ldk;
  p.v += bl * pin.kv_brows * p.ldv;
  if (p.gate) p.gate += bl * N * p.ldg;
  p.out += bl * N * p.ldo;
  p.prob += bl * 225 * N;
  return p;
}

#define LGP_DPPF(acc, N) "v_fmac_f32_dpp " acc ", %[t], %[x] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void lgp_dpp_axpy15(float (&s)[15], float t, float x) {
  asm("s_nop 1\n\t"
      LGP_DPPF("%0", 0) LGP_DPPF("%1", 1) LGP_DPPF("%2", 2) LGP_DPPF("%3", 3) LGP_DPPF("%4", 4) LGP_DPPF("%5", 5)
      LGP_DPPF("%6", 6) LGP_DPPF("%7", 7) LGP_DPPF("%8", 8) LGP_DPPF("%9", 9) LGP_DPPF("%10", 10) LGP_DPPF("%11", 11)
      LGP_DPPF("%12", 12) LGP_DPPF("%13", 13) LGP_DPPF("%14", 14)
      : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]), "+v"(s[8]),
        "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14])
      : [t] "v"(t), [x] "v"(x));
}
#undef LGP_DPPF

constexpr int LGP_R = 7, LGP_WS = 15, LGP_NPOS = 64 + 2 * LGP_R, LGP_LD = 36;
constexpr int LGP_NF4 = LGP_NPOS * 8, LGP_PER = (LGP_NF4 + 63) / 64, LGP_SLAB = (LGP_NPOS + 2) * LGP_LD;

// stage 32 channels [c0, c0+32) of image row ky, key positions x0-R .. x0+63+R, into a wave-private LDS slab
__device__ __forceinline__ void lgp_stage(const float* base, int ld, int c0, int ky, int x0, int w, int lane, float* slab) {
  float4 st[LGP_PER];
#pragma unroll
  for (int i = 0; i < LGP_PER; ++i) {
    const int f = lane + i * 64;
    const int pos = f >> 3, c4 = f & 7;
    const int kx = x0 - LGP_R + pos;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < LGP_NF4 && kx >= 0 && kx < w) t = *reinterpret_cast<const float4*>(base + ((long)ky * w + kx) * ld + c0 + c4 * 4);
    st[i] = t;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < LGP_PER; ++i) {
    const int f = lane + i * 64;
    if (f < LGP_NF4) *reinterpret_cast<float4*>(&slab[(f >> 3) * LGP_LD + (f & 7) * 4]) = st[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One workgroup per (64-wide strip of an image row, window row dy): its four waves take the four 32-channel chunks of the
// 128-wide q.k contraction and meet in LDS (summed in chunk order: deterministic).  (Round 2 gave a workgroup a whole image
// row and let eight waves walk the 15 window rows: 31 workgroups for a 480p map, 12 % of the CUs, 56 us; the window rows are
// independent outputs, so they are grid work: 465 workgroups.)
__global__ void __launch_bounds__(256) lgp_scores_kernel(const LgpParams pin) {
  __shared__ __attribute__((aligned(16))) float lds[4 * LGP_SLAB];
  __shared__ float part[3][LGP_WS][64];
  const int bl = blockIdx.z / LGP_WS, dy = blockIdx.z - bl * LGP_WS;
  const LgpParams p = lgp_lane(pin, bl);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x0 = blockIdx.x * 64, y = blockIdx.y;
  const bool active = x0 + lane < p.w;
  const int x = active ? x0 + lane : p.w - 1;
  const int n = y * p.w + x;
  const int N = p.h * p.w;
  const int ky = y + dy - LGP_R;
  if (ky < 0 || ky >= p.h) {           // whole window row outside the image (uniform over the workgroup)
    if (wave == 0 && active)
#pragma unroll
      for (int dx = 0; dx < LGP_WS; ++dx) p.prob[(long)(dy * LGP_WS + dx) * N + n] = -INFINITY;
    return;
  }
  float* slab = lds + wave * LGP_SLAB;
  const int c0 = wave * 32;
  float s[LGP_WS];
#pragma unroll
  for (int dx = 0; dx < LGP_WS; ++dx) s[dx] = 0.f;
  {
    float qs[32], tk[32];
    {
      const float4* src = reinterpret_cast<const float4*>(p.q + (long)n * p.ldq + c0);
      const float* wk = p.relk_t + ((long)dy * 128 + c0) * 16 + (lane & 15);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 t = src[i];
        qs[4 * i] = t.x / p.scale_div; qs[4 * i + 1] = t.y / p.scale_div;
        qs[4 * i + 2] = t.z / p.scale_div; qs[4 * i + 3] = t.w / p.scale_div;
      }
#pragma unroll
      for (int c = 0; c < 32; ++c) tk[c] = __builtin_nontemporal_load(wk + c * 16);
    }
    lgp_stage(p.k, p.ldk, c0, ky, x0, p.w, lane, slab);
#pragma unroll
    for (int dx = 0; dx < LGP_WS; ++dx) {
      float dot = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const float4 kk = *reinterpret_cast<const float4*>(&slab[(lane + dx) * LGP_LD + c4 * 4]);
        dot = fmaf(qs[4 * c4], kk.x, dot);
        dot = fmaf(qs[4 * c4 + 1], kk.y, dot);
        dot = fmaf(qs[4 * c4 + 2], kk.z, dot);
        dot = fmaf(qs[4 * c4 + 3], kk.w, dot);
      }
      s[dx] = dot;
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) lgp_dpp_axpy15(s, tk[c], qs[c]);
  }
  if (wave > 0) {
#pragma unroll
    for (int dx = 0; dx < LGP_WS; ++dx) part[wave - 1][dx][lane] = s[dx];
  }
  __syncthreads();
  if (wave == 0 && active) {
#pragma unroll
    for (int dx = 0; dx < LGP_WS; ++dx) {
      const float t = (((p.relk_b[dy * 16 + dx] + s[dx]) + part[0][dx][lane]) + part[1][dx][lane]) + part[2][dx][lane];
      const int kx = x + dx - LGP_R;
      p.prob[(long)(dy * LGP_WS + dx) * N + n] = (kx >= 0 && kx < p.w) ? t : -INFINITY;
    }
  }
}

// 16 queries x 16 slot groups per workgroup: thread (g, i) keeps slots g, g+16, ... of query n0+i in registers (one read,
// one write of P), the 16 partial maxima / sums of a query meet in LDS.  (One thread per query walked the 225 slots three
// times through a dependent load chain: 121 us for a 1.5 MB map.)
__global__ void __launch_bounds__(256) lgp_softmax_kernel(float* __restrict__ prob_all, int N) {
  constexpr int W2 = LGP_WS * LGP_WS, G = 16, PER = (W2 + G - 1) / G;
  float* __restrict__ prob = prob_all + (long)blockIdx.y * W2 * N;
  __shared__ float red[G][16];
  const int i = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int n = blockIdx.x * 16 + i;
  const bool ok = n < N;
  float e[PER];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int w = g + k * G;
    e[k] = (ok && w < W2) ? prob[(long)w * N + n] : -INFINITY;
    m = fmaxf(m, e[k]);
  }
  red[g][i] = m;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < G; ++k) m = fmaxf(m, red[k][i]);
  __syncthreads();
  float l = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    e[k] = expf(e[k] - m);          // exp(-inf) = 0 for slots outside the image
    l += e[k];
  }
  red[g][i] = l;
  __syncthreads();
  l = 0.f;
#pragma unroll
  for (int k = 0; k < G; ++k) l += red[k][i];     // fixed order: deterministic
  const float inv = 1.f / l;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int w = g + k * G;
    if (ok && w < W2) prob[(long)w * N + n] = e[k] * inv;
  }
}

// CB = channels of the value a workgroup aggregates (32 or 16).  The wave-private slab is (64 + 2 R + 2) x (CB + 4) floats: with CB = 32
// the eight slabs of a workgroup are 92 KB, ONE workgroup per CU and two waves per SIMD -- every exposed load of the stage -> compute
// chain idles the SIMD; CB = 16 (51 KB) lets three workgroups share a CU, CB = 8 (31 KB) five: 75.3 -> 71.7 -> 60.1 us for the three
// launches of a 480p frame (profiles/r06_lgp_cb.txt).  The sums per output are the same in the same order: bit-identical.
#ifndef AOT_LGP_CB
#define AOT_LGP_CB 8
#endif
template <int NWV, int CB>
__global__ void __launch_bounds__(NWV * 64) lgp_aggregate_kernel(const LgpParams pin) {
  constexpr int LD = CB + 4, C4 = CB / 4;
  constexpr int NF4 = LGP_NPOS * C4, PER = (NF4 + 63) / 64, SLAB = (LGP_NPOS + 2) * LD;
  static_assert(SLAB >= CB * 64, "the slab doubles as the cross-wave reduction buffer");
  __shared__ __attribute__((aligned(16))) float lds[NWV * SLAB];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bl = blockIdx.y / pin.h;
  const LgpParams p = lgp_lane(pin, bl);
  const int x0 = blockIdx.x * 64, y = blockIdx.y - bl * pin.h, c0 = blockIdx.z * CB;
  const bool active = x0 + lane < p.w;
  const int x = active ? x0 + lane : p.w - 1;
  const int n = y * p.w + x;
  const int N = p.h * p.w;
  float* slab = lds + wave * SLAB;
  float o[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) o[c] = 0.f;
  for (int dy = wave; dy < LGP_WS; dy += NWV) {
    const int ky = y + dy - LGP_R;
    if (ky < 0 || ky >= p.h) continue;
    float pw[LGP_WS];
#pragma unroll
    for (int dx = 0; dx < LGP_WS; ++dx) pw[dx] = p.prob[(long)(dy * LGP_WS + dx) * N + n];
    {          // stage CB channels [c0, c0 + CB) of image row ky, positions x0 - R .. x0 + 63 + R, into the wave's slab
      float4 st[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int f = lane + i * 64;
        const int pos = f / C4, c4 = f - pos * C4;
        const int kx = x0 - LGP_R + pos;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < NF4 && kx >= 0 && kx < p.w) t = *reinterpret_cast<const float4*>(p.v + ((long)ky * p.w + kx) * p.ldv + c0 + c4 * 4);
        st[i] = t;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int f = lane + i * 64;
        if (f < NF4) *reinterpret_cast<float4*>(&slab[(f / C4) * LD + (f % C4) * 4]) = st[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int dx = 0; dx < LGP_WS; ++dx) {
#pragma unroll
      for (int c4 = 0; c4 < C4; ++c4) {
        const float4 vv = *reinterpret_cast<const float4*>(&slab[(lane + dx) * LD + c4 * 4]);
        o[4 * c4] = fmaf(pw[dx], vv.x, o[4 * c4]);
        o[4 * c4 + 1] = fmaf(pw[dx], vv.y, o[4 * c4 + 1]);
        o[4 * c4 + 2] = fmaf(pw[dx], vv.z, o[4 * c4 + 2]);
        o[4 * c4 + 3] = fmaf(pw[dx], vv.w, o[4 * c4 + 3]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // sum the NWV partials in fixed order
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CB; ++c) slab[c * 64 + lane] = o[c];
  __syncthreads();
  if (wave == 0 && active) {
    float t[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) t[c] = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NWV; ++w2)
#pragma unroll
      for (int c = 0; c < CB; ++c) t[c] += lds[w2 * SLAB + c * 64 + lane];
    float4* dst = reinterpret_cast<float4*>(p.out + (long)n * p.ldo + c0);
    const float4* g = p.gate ? reinterpret_cast<const float4*>(p.gate + (long)n * p.ldg + c0) : nullptr;
#pragma unroll
    for (int c4 = 0; c4 < C4; ++c4) {
      float4 r = make_float4(t[4 * c4], t[4 * c4 + 1], t[4 * c4 + 2], t[4 * c4 + 3]);
      if (g) { const float4 u = g[c4]; r.x *= u.x; r.y *= u.y; r.z *= u.z; r.w *= u.w; }
      dst[c4] = r;
    }
  }
}

extern "C" int aot_local_gated_f32(const float* q, const float* k, const float* v, const float* gate, const float* relk_t,
                                   const float* relk_b, float* prob, float* out, int B, long kv_brows, int h, int w,
                                   int dqk, int dv, int max_dis, int ldq, int ldk, int ldv, int ldg, int ldo,
                                   float scale_div, void* stream) {
  if (!q || !k || !v || !relk_t || !relk_b || !prob || !out || h <= 0 || w <= 0 || B <= 0) return AOT_ERR_BADARG;
  if (B > 1 && kv_brows < (long)h * w) return AOT_ERR_BADARG;
  if ((long)B * h > 65535 || (long)B * LGP_WS > 65535) return AOT_ERR_UNSUPPORTED;
  if (dqk != 128 || max_dis != 7 || dv <= 0 || (dv & 31)) return AOT_ERR_UNSUPPORTED;
  if ((ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3) || (gate && (ldg & 3))) return AOT_ERR_BADARG;
  LgpParams p;
  p.q = q; p.k = k; p.v = v; p.gate = gate; p.relk_t = relk_t; p.relk_b = relk_b; p.prob = prob; p.out = out;
  p.h = h; p.w = w; p.C = dv; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldg = ldg; p.ldo = ldo; p.scale_div = scale_div;
  p.kv_brows = kv_brows;
  hipStream_t s = (hipStream_t)stream;
  constexpr int NWV = 8;
  hipLaunchKernelGGL(lgp_scores_kernel, dim3(cdiv(w, 64), h, B * LGP_WS), dim3(256), 0, s, p);
  hipLaunchKernelGGL(lgp_softmax_kernel, dim3(cdiv(h * w, 16), B), dim3(256), 0, s, prob, h * w);
  hipLaunchKernelGGL((lgp_aggregate_kernel<NWV, AOT_LGP_CB>), dim3(cdiv(w, 64), B * h, dv / AOT_LGP_CB), dim3(NWV * 64), 0, s, p);
  AOT_LAUNCH_CHECK();
}
