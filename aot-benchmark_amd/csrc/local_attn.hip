// Short-term (windowed) attention of AOT, one fused kernel.
//
// For query p = (y, x) of head hd, over the (2R+1)^2 window slots w = (dy, dx):
//   s_w  = (q/sqrt(d)) . k[p + delta(w)]  +  ( relk_w[hd, w, :] . q  + relk_b[hd, w] )     (rel on UNSCALED q)
//   a    = softmax_w(s)  over in-image slots (the reference pushes the others to -1e8 -> exactly 0)
//   out  = sum_w a_w * ( v[p + delta(w)] + relv[hd, :, w] )
// This is what the reference computes with the CUDA correlation sampler (or a 386 MB unfold), a 164 MB
// scatter to a dense N x N map and a dense matmul (attention.py:308-428); here nothing is materialised.
//
// Mapping: a workgroup = one 64-lane wave = 64 consecutive queries of one image row and one head; lane =
// query.  The window is walked row by row (dy): the K and V rows y+dy-R (64 + 2R positions x 32 channels)
// are staged in LDS token-major with a 36-float stride, so each lane's sliding 15-key window is read
// with conflict-free ds_read_b128 (stride 36 dwords: 16 lanes -> 16 distinct 4-bank slots).  Scores of one
// window row (15 per lane) are combined with an online softmax; the relative-position tables are
// wave-uniform and come in through the scalar cache.  The work is 2 x 0.39 GFLOP/layer of irregular
// dot products -- VALU work by nature (fp32 MFMA has the same peak and would waste 3/4 of a dense tile).
#include "common.h"

struct LocalParams {
  const float* q;
  const float* k;
  const float* v;
  const float* relk_w;  // [H*W2][32]
  const float* relk_b;  // [H*W2]
  const float* relv_t;  // [H][W2][32]
  float* out;
  int h, w, H, ldq, ldk, ldv, ldo;
  float scale_div;
};

template <int R>
__global__ void __launch_bounds__(64) local_attn_d32_kernel(const LocalParams p) {
  constexpr int WS = 2 * R + 1, W2 = WS * WS, D = 32;
  constexpr int NPOS = 64 + 2 * R;  // staged key positions per row
  constexpr int LDS_LD = 36;
  __shared__ __attribute__((aligned(16))) float Ks[NPOS + 2][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Vs[NPOS + 2][LDS_LD];

  const int lane = threadIdx.x;
  const int x0 = blockIdx.x * 64, y = blockIdx.y, hd = blockIdx.z;
  const int x = x0 + lane;
  const bool active = x < p.w;
  const int n = y * p.w + (active ? x : p.w - 1);

  float qu[D], qs[D];
  {
    const float4* src = reinterpret_cast<const float4*>(p.q + (long)n * p.ldq + hd * D);
#pragma unroll
    for (int i = 0; i < D / 4; ++i) {
      const float4 t = src[i];
      qu[4 * i] = t.x; qu[4 * i + 1] = t.y; qu[4 * i + 2] = t.z; qu[4 * i + 3] = t.w;
    }
#pragma unroll
    for (int c = 0; c < D; ++c) qs[c] = qu[c] / p.scale_div;
  }

  float m = -INFINITY, l = 0.f;
  float o[D];
#pragma unroll
  for (int c = 0; c < D; ++c) o[c] = 0.f;

  const float* relk_w = p.relk_w + (long)hd * W2 * D;
  const float* relk_b = p.relk_b + hd * W2;
  const float* relv = p.relv_t + (long)hd * W2 * D;

  for (int dy = 0; dy < WS; ++dy) {
    const int ky = y + dy - R;
    if (ky < 0 || ky >= p.h) continue;  // wave-uniform: whole window row outside the image
    __syncthreads();
    // stage K and V row ky, positions kx = x0 - R + pos, zero outside the image
    for (int f = lane; f < NPOS * (D / 4); f += 64) {
      const int pos = f >> 3, c4 = f & 7;
      const int kx = x0 - R + pos;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (kx >= 0 && kx < p.w) {
        const long tok = (long)ky * p.w + kx;
        kv = *reinterpret_cast<const float4*>(p.k + tok * p.ldk + hd * D + c4 * 4);
        vv = *reinterpret_cast<const float4*>(p.v + tok * p.ldv + hd * D + c4 * 4);
      }
      *reinterpret_cast<float4*>(&Ks[pos][c4 * 4]) = kv;
      *reinterpret_cast<float4*>(&Vs[pos][c4 * 4]) = vv;
    }
    __syncthreads();
    if (!active) continue;

    float s[WS];
#pragma unroll
    for (int dx = 0; dx < WS; ++dx) {
      const int wi = dy * WS + dx;
      float dot = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < D / 4; ++c4) {
        const float4 kk = *reinterpret_cast<const float4*>(&Ks[lane + dx][c4 * 4]);
        dot = fmaf(qs[4 * c4], kk.x, dot);
        dot = fmaf(qs[4 * c4 + 1], kk.y, dot);
        dot = fmaf(qs[4 * c4 + 2], kk.z, dot);
        dot = fmaf(qs[4 * c4 + 3], kk.w, dot);
      }
      float rel = 0.f;
      const float* wk = relk_w + wi * D;
#pragma unroll
      for (int c = 0; c < D; ++c) rel = fmaf(qu[c], wk[c], rel);
      rel += relk_b[wi];
      const int kx = x + dx - R;
      s[dx] = (kx >= 0 && kx < p.w) ? dot + rel : -INFINITY;
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
      const float pw = expf(s[dx] - mnew);  // exp(-inf) = 0 for masked slots
      l += pw;
      const float* rv = relv + (dy * WS + dx) * D;
#pragma unroll
      for (int c4 = 0; c4 < D / 4; ++c4) {
        const float4 vv = *reinterpret_cast<const float4*>(&Vs[lane + dx][c4 * 4]);
        o[4 * c4] = fmaf(pw, vv.x + rv[4 * c4], o[4 * c4]);
        o[4 * c4 + 1] = fmaf(pw, vv.y + rv[4 * c4 + 1], o[4 * c4 + 1]);
        o[4 * c4 + 2] = fmaf(pw, vv.z + rv[4 * c4 + 2], o[4 * c4 + 2]);
        o[4 * c4 + 3] = fmaf(pw, vv.w + rv[4 * c4 + 3], o[4 * c4 + 3]);
      }
    }
  }

  if (active) {
    const float inv = 1.f / l;
    float4* dst = reinterpret_cast<float4*>(p.out + (long)n * p.ldo + hd * D);
#pragma unroll
    for (int c4 = 0; c4 < D / 4; ++c4)
      dst[c4] = make_float4(o[4 * c4] * inv, o[4 * c4 + 1] * inv, o[4 * c4 + 2] * inv, o[4 * c4 + 3] * inv);
  }
}

extern "C" int aot_local_attn_f32(const float* q, const float* k, const float* v, const float* relk_w,
                                  const float* relk_b, const float* relv_t, float* out, int h, int w, int H,
                                  int d, int max_dis, int ldq, int ldk, int ldv, int ldo, float scale_div,
                                  void* stream) {
  if (!q || !k || !v || !relk_w || !relk_b || !relv_t || !out || h <= 0 || w <= 0 || H <= 0) return AOT_ERR_BADARG;
  if (d != 32 || max_dis != 7) return AOT_ERR_UNSUPPORTED;
  if ((ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3)) return AOT_ERR_BADARG;
  LocalParams p;
  p.q = q; p.k = k; p.v = v; p.relk_w = relk_w; p.relk_b = relk_b; p.relv_t = relv_t; p.out = out;
  p.h = h; p.w = w; p.H = H; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.scale_div = scale_div;
  hipLaunchKernelGGL(local_attn_d32_kernel<7>, dim3(cdiv(w, 64), h, H), dim3(64), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}
