// Top-k sparse long-term attention (reference MultiheadAttention with top_k > 0, attention.py:102-105):
//   scores = (Q/sqrt(d)) K^T per head;  keep the top_k scores of every query row;  softmax over those;  P V.
// A default-off evaluation knob of the reference (long videos), so this path favours exactness and simplicity over
// speed: the scores are materialised once on the matrix cores, a radix select finds each row's k-th largest score, and
// the few selected keys are gathered (top_k x 32 FMAs per row and head instead of a dense P.V).
#include "common.h"

struct TopkParams {
  const float* q;
  const float* k;
  const float* v;
  float* out;
  float* scores;   // [H][Nq][ldS]
  int Nq, T, H, ldq, ldk, ldv, ldo, ldS, top_k;
  float scale_div;
};

// S[h][q][t]: one wave = 32 queries x 256 keys of one head.  Query on the A side, so a score register holds 32
// consecutive keys of one query row across the lanes -> 128-byte coalesced stores.
__global__ void __launch_bounds__(64) attn_scores_kernel(const TopkParams p) {
  const int h = blockIdx.x, kb = blockIdx.y, qt = blockIdx.z;
  const int lane = threadIdx.x, j = lane & 31, hi = lane >> 5;
  float qf[16];
  {
    const int qrow = min(qt * 32 + j, p.Nq - 1);
    const float4* src = reinterpret_cast<const float4*>(p.q + (long)qrow * p.ldq + h * 32 + hi * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 t = src[i];
      qf[4 * i + 0] = t.x / p.scale_div;
      qf[4 * i + 1] = t.y / p.scale_div;
      qf[4 * i + 2] = t.z / p.scale_div;
      qf[4 * i + 3] = t.w / p.scale_div;
    }
  }
  const float* kptr = p.k + h * 32 + hi * 16;
  for (int kt = kb * 256; kt < min(p.T, kb * 256 + 256); kt += 32) {
    float kf[16];
    const float4* src = reinterpret_cast<const float4*>(kptr + (long)min(kt + j, p.T - 1) * p.ldk);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 t = src[i];
      kf[4 * i] = t.x; kf[4 * i + 1] = t.y; kf[4 * i + 2] = t.z; kf[4 * i + 3] = t.w;
    }
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[s], kf[s], sc, 0, 0, 0);
    if (kt + j < p.T) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = qt * 32 + mfma32_row(r, hi);
        if (qi < p.Nq) p.scores[((long)h * p.Nq + qi) * p.ldS + kt + j] = sc[r];
      }
    }
  }
}

// order-preserving map float -> uint (larger float <=> larger uint)
__device__ __forceinline__ unsigned ord_u32(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// One 256-thread block per (query, head) row: radix select of the k-th largest score (4 passes of 8 bits), then
// softmax over the selected keys and the gather-accumulate of their V rows.  Exactly top_k keys are used; among
// scores EQUAL to the k-th largest the choice is arbitrary (as in torch.topk).
__global__ void __launch_bounds__(256) attn_topk_gather_kernel(const TopkParams p) {
  const int qi = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const float* s = p.scores + ((long)h * p.Nq + qi) * p.ldS;
  __shared__ unsigned hist[256];
  __shared__ unsigned sh_prefix, sh_need, sh_eq;
  __shared__ float red[256][33];
  unsigned prefix = 0, mask = 0, need = (unsigned)p.top_k;
  float mx = -INFINITY;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    for (int t = tid; t < p.T; t += 256) {
      const float f = s[t];
      if (pass == 0) mx = fmaxf(mx, f);
      const unsigned u = ord_u32(f);
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned cum = 0;
      int b = 255;
      for (; b > 0; --b) {
        if (cum + hist[b] >= need) break;
        cum += hist[b];
      }
      sh_prefix = prefix | ((unsigned)b << shift);
      sh_need = need - cum;       // how many of the elements inside bin b are still wanted
      sh_eq = 0;
    }
    __syncthreads();
    prefix = sh_prefix;
    need = sh_need;
    mask |= 0xFFu << shift;
    __syncthreads();
  }
  // row maximum (the top-1 score) for the stable softmax
  red[tid][0] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid][0] = fmaxf(red[tid][0], red[tid + o][0]);
    __syncthreads();
  }
  mx = red[0][0];
  __syncthreads();
  const unsigned thr = prefix;     // ord of the k-th largest score; `need` of the scores equal to it are taken
  float acc[32];
#pragma unroll
  for (int d = 0; d < 32; ++d) acc[d] = 0.f;
  float l = 0.f;
  for (int t = tid; t < p.T; t += 256) {
    const float f = s[t];
    const unsigned u = ord_u32(f);
    bool take = u > thr;
    if (u == thr) take = atomicAdd(&sh_eq, 1u) < need;
    if (take) {
      const float w = expf(f - mx);
      l += w;
      const float4* vr = reinterpret_cast<const float4*>(p.v + (long)t * p.ldv + h * 32);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 x = vr[i];
        acc[4 * i] += w * x.x; acc[4 * i + 1] += w * x.y; acc[4 * i + 2] += w * x.z; acc[4 * i + 3] += w * x.w;
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 32; ++d) red[tid][d] = acc[d];
  red[tid][32] = l;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
#pragma unroll
      for (int d = 0; d < 33; ++d) red[tid][d] += red[tid + o][d];
    }
    __syncthreads();
  }
  if (tid < 32) p.out[(long)qi * p.ldo + h * 32 + tid] = red[0][tid] / red[0][32];
}

extern "C" int aot_attn_topk_f32(const float* q, const float* k, const float* v, float* out, float* scores, int Nq, int T,
                                 int H, int d, int ldq, int ldk, int ldv, int ldo, float scale_div, int top_k,
                                 void* stream) {
  if (d != 32) return AOT_ERR_UNSUPPORTED;
  if (!q || !k || !v || !out || !scores || Nq <= 0 || T <= 0 || H <= 0) return AOT_ERR_BADARG;
  if (top_k <= 0 || top_k >= T) return AOT_ERR_BADARG;      // top_k >= T is the dense softmax: use aot_attn_f32
  if ((ldq & 3) || (ldk & 3) || (ldv & 3) || ((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15))
    return AOT_ERR_BADARG;
  TopkParams p;
  p.q = q; p.k = k; p.v = v; p.out = out; p.scores = scores;
  p.Nq = Nq; p.T = T; p.H = H; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.ldS = (T + 3) & ~3;
  p.top_k = top_k; p.scale_div = scale_div;
  hipLaunchKernelGGL(attn_scores_kernel, dim3(H, cdiv(T, 256), cdiv(Nq, 32)), dim3(64), 0, (hipStream_t)stream, p);
  hipLaunchKernelGGL(attn_topk_gather_kernel, dim3(Nq, H), dim3(256), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// Gated form (DeAOT GatedPropagation with top_k > 0, attention.py:689-693): ONE head of width d = 32 * DCH (128) and a wide
// value (dv = 1024 = [V | ID_V]); out = (sum over the top_k keys of softmax weight x V row) * gate.
// ---------------------------------------------------------------------------------------------------------------------
struct GTopkParams {
  const float* q;
  const float* k;
  const float* v;
  const float* gate;
  float* out;
  float* scores;   // [Nq][ldS]
  int Nq, T, dv, ldq, ldk, ldv, ldg, ldo, ldS, top_k;
  float scale_div;
};

// S[q][t] for one head of width 32 * DCH: one wave = 32 queries x 256 keys, the contraction walked in 32-wide chunks
template <int DCH>
__global__ void __launch_bounds__(64) gattn_scores_kernel(const GTopkParams p) {
  const int kb = blockIdx.x, qt = blockIdx.y;
  const int lane = threadIdx.x, j = lane & 31, hi = lane >> 5;
  float qf[DCH][16];
  {
    const int qrow = min(qt * 32 + j, p.Nq - 1);
#pragma unroll
    for (int c = 0; c < DCH; ++c) {
      const float4* src = reinterpret_cast<const float4*>(p.q + (long)qrow * p.ldq + c * 32 + hi * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 t = src[i];
        qf[c][4 * i + 0] = t.x / p.scale_div;
        qf[c][4 * i + 1] = t.y / p.scale_div;
        qf[c][4 * i + 2] = t.z / p.scale_div;
        qf[c][4 * i + 3] = t.w / p.scale_div;
      }
    }
  }
  for (int kt = kb * 256; kt < min(p.T, kb * 256 + 256); kt += 32) {
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
    const float* krow = p.k + (long)min(kt + j, p.T - 1) * p.ldk + hi * 16;
#pragma unroll
    for (int c = 0; c < DCH; ++c) {
      float kf[16];
      const float4* src = reinterpret_cast<const float4*>(krow + c * 32);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 t = src[i];
        kf[4 * i] = t.x; kf[4 * i + 1] = t.y; kf[4 * i + 2] = t.z; kf[4 * i + 3] = t.w;
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[c][s], kf[s], sc, 0, 0, 0);
    }
    if (kt + j < p.T) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = qt * 32 + mfma32_row(r, hi);
        if (qi < p.Nq) p.scores[(long)qi * p.ldS + kt + j] = sc[r];
      }
    }
  }
}

// Radix select of the k-th largest score of one row by a 256-thread block (4 passes of 8 bits): returns the order key of
// that score, how many of the scores EQUAL to it are still wanted, and the row maximum.
__device__ __forceinline__ void select_kth(const float* s, int T, int top_k, unsigned* hist, unsigned* sh, unsigned& thr,
                                           unsigned& need, float& mx, float* redmax) {
  const int tid = threadIdx.x;
  unsigned prefix = 0, mask = 0;
  need = (unsigned)top_k;
  mx = -INFINITY;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
      const float f = s[t];
      if (pass == 0) mx = fmaxf(mx, f);
      const unsigned u = ord_u32(f);
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned cum = 0;
      int b = 255;
      for (; b > 0; --b) {
        if (cum + hist[b] >= need) break;
        cum += hist[b];
      }
      sh[0] = prefix | ((unsigned)b << shift);
      sh[1] = need - cum;
    }
    __syncthreads();
    prefix = sh[0];
    need = sh[1];
    mask |= 0xFFu << shift;
    __syncthreads();
  }
  redmax[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) redmax[tid] = fmaxf(redmax[tid], redmax[tid + o]);
    __syncthreads();
  }
  mx = redmax[0];
  __syncthreads();
  thr = prefix;
}

// One 256-thread block per query row.  After the select, the row is walked once more in key order: the selected keys are
// compacted IN ORDER into an LDS list (ballot prefix sums; among scores equal to the k-th largest the first `need` in key
// order are taken -- deterministic, torch.topk leaves that choice open), and every thread accumulates its own four (x NV)
// channels of the selected V rows -- coalesced 4 KB row reads, identical summation order on every run.
template <int NV>
__global__ void __launch_bounds__(256) gattn_topk_gather_kernel(const GTopkParams p) {
  constexpr int CAP = 1024;
  const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* s = p.scores + (long)qi * p.ldS;
  __shared__ unsigned hist[256];
  __shared__ unsigned sh[2];
  __shared__ float redmax[256];
  __shared__ int sel_t[CAP];
  __shared__ float sel_w[CAP];
  __shared__ unsigned wcnt[2][4];
  unsigned thr, need;
  float mx;
  select_kth(s, p.T, p.top_k, hist, sh, thr, need, mx, redmax);
  float acc[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = 0.f;
  float l = 0.f;
  unsigned count = 0, eq_seen = 0;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  auto flush = [&]() {
    __syncthreads();
    for (unsigned i = 0; i < count; ++i) {
      const float w = sel_w[i];
      l += w;
      const float* vr = p.v + (long)sel_t[i] * p.ldv;
#pragma unroll
      for (int g = 0; g < NV; ++g) {
        const int c4 = (g * 256 + tid) * 4;
        if (c4 < p.dv) {
          const float4 x = *reinterpret_cast<const float4*>(vr + c4);
          acc[g][0] += w * x.x; acc[g][1] += w * x.y; acc[g][2] += w * x.z; acc[g][3] += w * x.w;
        }
      }
    }
    count = 0;
    __syncthreads();
  };
  for (int base = 0; base < p.T; base += 256) {
    const int t = base + tid;
    const bool in = t < p.T;
    const float f = in ? s[t] : -INFINITY;
    const unsigned u = ord_u32(f);
    const bool gt = in && u > thr, eq = in && u == thr;
    const unsigned long long beq = __ballot(eq);
    if (lane == 0) wcnt[0][wave] = (unsigned)__popcll(beq);
    __syncthreads();
    unsigned eq_rank = eq_seen + (unsigned)__popcll(beq & lt_mask), eq_total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      if (w2 < wave) eq_rank += wcnt[0][w2];
      eq_total += wcnt[0][w2];
    }
    const bool take = gt || (eq && eq_rank < need);
    const unsigned long long btk = __ballot(take);
    if (lane == 0) wcnt[1][wave] = (unsigned)__popcll(btk);
    __syncthreads();
    unsigned pos = count + (unsigned)__popcll(btk & lt_mask), tk_total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      if (w2 < wave) pos += wcnt[1][w2];
      tk_total += wcnt[1][w2];
    }
    if (take) {
      sel_t[pos] = t;
      sel_w[pos] = expf(f - mx);
    }
    count += tk_total;
    eq_seen += eq_total;
    if (count + 256 > CAP) flush();
  }
  flush();
#pragma unroll
  for (int g = 0; g < NV; ++g) {
    const int c4 = (g * 256 + tid) * 4;
    if (c4 < p.dv) {
      float4 gt4 = {1.f, 1.f, 1.f, 1.f};
      if (p.gate) gt4 = *reinterpret_cast<const float4*>(p.gate + (long)qi * p.ldg + c4);
      float4 o;
      o.x = acc[g][0] / l * gt4.x; o.y = acc[g][1] / l * gt4.y; o.z = acc[g][2] / l * gt4.z; o.w = acc[g][3] / l * gt4.w;
      *reinterpret_cast<float4*>(p.out + (long)qi * p.ldo + c4) = o;
    }
  }
}

extern "C" int aot_gated_attn_topk_f32(const float* q, const float* k, const float* v, const float* gate, float* out,
                                       float* scores, int Nq, int T, int d, int dv, int ldq, int ldk, int ldv, int ldg,
                                       int ldo, float scale_div, int top_k, void* stream) {
  if (d != 128 || dv <= 0 || (dv & 3) || dv > 2048) return AOT_ERR_UNSUPPORTED;
  if (!q || !k || !v || !out || !scores || Nq <= 0 || T <= 0) return AOT_ERR_BADARG;
  if (top_k <= 0 || top_k >= T) return AOT_ERR_BADARG;      // top_k >= T is the dense softmax: use aot_gated_attn_f32
  if ((ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3) || (gate && (ldg & 3)) || ((uintptr_t)q & 15) || ((uintptr_t)k & 15) ||
      ((uintptr_t)v & 15) || ((uintptr_t)out & 15) || (gate && ((uintptr_t)gate & 15)))
    return AOT_ERR_BADARG;
  GTopkParams p;
  p.q = q; p.k = k; p.v = v; p.gate = gate; p.out = out; p.scores = scores;
  p.Nq = Nq; p.T = T; p.dv = dv; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldg = ldg; p.ldo = ldo; p.ldS = (T + 3) & ~3;
  p.top_k = top_k; p.scale_div = scale_div;
  hipLaunchKernelGGL(gattn_scores_kernel<4>, dim3(cdiv(T, 256), cdiv(Nq, 32)), dim3(64), 0, (hipStream_t)stream, p);
  if (dv <= 1024)
    hipLaunchKernelGGL(gattn_topk_gather_kernel<1>, dim3(Nq), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(gattn_topk_gather_kernel<2>, dim3(Nq), dim3(256), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}
