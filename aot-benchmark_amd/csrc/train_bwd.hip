// Training path: the primitives the differentiable forward of the training engine is built from, with their
// backward kernels.  What autograd derives for the reference's networks/engines/aot_engine.py:33-108 (loss.backward(),
// networks/managers/trainer.py:460-519) decomposes into a handful of linear maps and pointwise / normalisation rules:
//
//   aot_matmul_strided_f32      C[b] = alpha * A[b] . B[b] (+ bias) for operands with ARBITRARY element strides: nn.Linear and
//                               1x1 convs (transformer.py:321-359, fpn.py:34-58, mobilenetv2.py), the QK^T / PV products of
//                               MultiheadAttention / GatedPropagation (attention.py:92-117, 672-707) per head, and -- with
//                               the strides of a transposed view -- every one of their gradients (dX = dY W, dW = dY^T X,
//                               dV = P^T dO, dP = dO V^T, ...): ONE kernel, exact k-ordered fp32 fma chains on
//                               v_mfma_f32_32x32x2_f32
//   aot_im2col_f32 / aot_col2im_f32      KxK convolutions (fpn.py 3x3, the identity bank's 17x17 / stride 16, aot.py:50-63) as
//                               im2col + matmul; col2im is the adjoint (gather form: deterministic)
//   aot_dwconv2d_bwd_data_f32 / aot_dwconv2d_bwd_weight_f32    depthwise KxK (basic.py:19-25,41-47, mobilenetv2.py:93-98)
//   aot_act_f32 / aot_act_bwd_f32        ReLU / ReLU6 / exact-erf GELU / SiLU
//   aot_layernorm_bwd_f32, aot_groupnorm_bwd_f32 (+ aot_norm_param_grads_f32)   nn.LayerNorm, nn.GroupNorm
//   aot_softmax_rows_f32 / aot_softmax_rows_bwd_f32            softmax over the keys / the 225 window slots
//   aot_bilinear_bwd_nhwc_f32            adjoint of aot_bilinear_nhwc_f32 (fpn.py:44-55, aot_engine.py:372-378)
//   aot_window_gather_f32 / aot_window_scatter_f32             local2global of the windowed attentions (attention.py:378-417,
//                               863-903) and its adjoint: dense [N, N] <-> window-slot [N, 225] layouts
//
// Correctness first: these are streaming / direct kernels with fixed summation order (deterministic), not tuned ones; the
// inference path does not use them.
#include "common.h"
#include <cmath>

namespace {

// ---- strided batched matmul -------------------------------------------------------------------------------------------
struct MatmulParams {
  const float* a;
  const float* b;
  const float* bias;   // [N] or null
  float* c;            // [batch][M][ldc]
  long sab, sam, sak;  // element strides of A[batch][m][k]
  long sbb, sbk, sbn;  // element strides of B[batch][k][n]
  long scb;            // batch stride of C (elements)
  int M, N, K, ldc;
  float alpha;
  int accumulate;      // C += ... instead of C = ...
};

// one wave = one 32x32 tile of C; a 256-thread workgroup = 64x64.  Operands straight from global memory (4-byte gathers with
// the caller's strides, zeros beyond the edges): general, not fast.
__global__ void __launch_bounds__(256) matmul_strided_kernel(const MatmulParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.y * 64 + (wave >> 1) * 32, n0 = blockIdx.x * 64 + (wave & 1) * 32;
  if (m0 >= p.M || n0 >= p.N) return;
  const int bt = blockIdx.z;
  const float* A = p.a + (long)bt * p.sab;
  const float* B = p.b + (long)bt * p.sbb;
  const int am = m0 + l31, bn = n0 + l31;
  const bool aok = am < p.M, bok = bn < p.N;
  const float* ap = A + (long)(aok ? am : 0) * p.sam;
  const float* bp = B + (long)(bok ? bn : 0) * p.sbn;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int k = 0;
  for (; k + 8 <= p.K; k += 8) {
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long kk = k + 2 * u + half;
      av[u] = aok ? ap[kk * p.sak] : 0.f;
      bv[u] = bok ? bp[kk * p.sbk] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
  }
  for (; k < p.K; k += 2) {
    const long kk = k + half;
    const bool kok = kk < p.K;
    const float av = (aok && kok) ? ap[kk * p.sak] : 0.f;
    const float bv = (bok && kok) ? bp[kk * p.sbk] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  }
  if (!bok) return;
  const float bias = p.bias ? p.bias[bn] : 0.f;
  float* C = p.c + (long)bt * p.scb;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + mfma32_row(r, half);
    if (m < p.M) {
      float* dst = C + (long)m * p.ldc + bn;
      const float v = p.alpha * acc[r] + bias;
      *dst = p.accumulate ? *dst + v : v;
    }
  }
}

// ---- im2col / col2im --------------------------------------------------------------------------------------------------
struct ColParams {
  int B, H, W, C, OH, OW, KH, KW, stride, pad, dil;
};

// cols[(b, oy, ox)][(ky, kx, c)] = x[(b, iy, ix)][c] or 0; one thread per 4 channels of one (output pixel, tap)
__global__ void __launch_bounds__(256) im2col_kernel(const float* __restrict__ x, float* __restrict__ cols, const ColParams p) {
  const int nv = p.C >> 2;
  const long total = (long)p.B * p.OH * p.OW * p.KH * p.KW * nv;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % nv);
  long t = idx / nv;
  const int tap = (int)(t % (p.KH * p.KW));
  t /= p.KH * p.KW;
  const int ox = (int)(t % p.OW);
  t /= p.OW;
  const int oy = (int)(t % p.OH);
  const int b = (int)(t / p.OH);
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int iy = oy * p.stride - p.pad + ky * p.dil, ix = ox * p.stride - p.pad + kx * p.dil;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
    v = *reinterpret_cast<const float4*>(x + (((long)b * p.H + iy) * p.W + ix) * p.C + c4 * 4);
  *reinterpret_cast<float4*>(cols + idx * 4) = v;
}

// dx[(b, iy, ix)][c] = sum over the taps (ky, kx) and output pixels that read this input pixel, in tap order
__global__ void __launch_bounds__(256) col2im_kernel(const float* __restrict__ cols, float* __restrict__ dx, const ColParams p) {
  const int nv = p.C >> 2;
  const long total = (long)p.B * p.H * p.W * nv;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % nv);
  long t = idx / nv;
  const int ix = (int)(t % p.W);
  t /= p.W;
  const int iy = (int)(t % p.H);
  const int b = (int)(t / p.H);
  const long krow = (long)p.KH * p.KW * p.C;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ky = 0; ky < p.KH; ++ky) {
    const int ny = iy + p.pad - ky * p.dil;
    if (ny < 0 || ny % p.stride) continue;
    const int oy = ny / p.stride;
    if (oy >= p.OH) continue;
    for (int kx = 0; kx < p.KW; ++kx) {
      const int nx = ix + p.pad - kx * p.dil;
      if (nx < 0 || nx % p.stride) continue;
      const int ox = nx / p.stride;
      if (ox >= p.OW) continue;
      const float4 v = *reinterpret_cast<const float4*>(cols + (((long)b * p.OH + oy) * p.OW + ox) * krow +
                                                        (long)(ky * p.KW + kx) * p.C + c4 * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  *reinterpret_cast<float4*>(dx + idx * 4) = acc;
}

// ---- depthwise convolution backward -----------------------------------------------------------------------------------
// dx[(b, iy, ix)][c] = sum_{ky,kx} dy[(b, oy, ox)][c] * w[(ky, kx)][c]   over the output pixels that read this input pixel
__global__ void __launch_bounds__(256) dwconv_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, const ColParams p) {
  const int nv = p.C >> 2;
  const long total = (long)p.B * p.H * p.W * nv;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % nv);
  long t = idx / nv;
  const int ix = (int)(t % p.W);
  t /= p.W;
  const int iy = (int)(t % p.H);
  const int b = (int)(t / p.H);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ky = 0; ky < p.KH; ++ky) {
    const int ny = iy + p.pad - ky * p.dil;
    if (ny < 0 || ny % p.stride) continue;
    const int oy = ny / p.stride;
    if (oy >= p.OH) continue;
    for (int kx = 0; kx < p.KW; ++kx) {
      const int nx = ix + p.pad - kx * p.dil;
      if (nx < 0 || nx % p.stride) continue;
      const int ox = nx / p.stride;
      if (ox >= p.OW) continue;
      const float4 g = *reinterpret_cast<const float4*>(dy + (((long)b * p.OH + oy) * p.OW + ox) * p.C + c4 * 4);
      const float4 k = *reinterpret_cast<const float4*>(w + (long)(ky * p.KW + kx) * p.C + c4 * 4);
      acc.x = fmaf(g.x, k.x, acc.x); acc.y = fmaf(g.y, k.y, acc.y);
      acc.z = fmaf(g.z, k.z, acc.z); acc.w = fmaf(g.w, k.w, acc.w);
    }
  }
  *reinterpret_cast<float4*>(dx + idx * 4) = acc;
}

// dw[(ky, kx)][c] = sum_{b, oy, ox} dy[(b, oy, ox)][c] * x[(b, iy, ix)][c]: one workgroup per (tap, 16 channels), its 16 thread rows
// split the output pixels; per-thread fp64 partials, fixed order (deterministic).  (Round 4 tried two levels -- workgroup = (pixel
// chunk, 64 channels), all 25 taps of a channel in registers, chunk partials summed by the last arriver: 152 us against this
// kernel's 128 us on the GPM's 5x5 convolutions, dropped.)
__global__ void __launch_bounds__(256) dwconv_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                float* __restrict__ dw, const ColParams p) {
  const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int tap = blockIdx.x, c = blockIdx.y * 16 + cl;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  __shared__ double red[16][16];
  double acc = 0.0;
  if (c < p.C) {
    const long npix = (long)p.B * p.OH * p.OW;
    for (long q = part; q < npix; q += 16) {
      const int ox = (int)(q % p.OW);
      long t = q / p.OW;
      const int oy = (int)(t % p.OH);
      const int b = (int)(t / p.OH);
      const int iy = oy * p.stride - p.pad + ky * p.dil, ix = ox * p.stride - p.pad + kx * p.dil;
      if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
      acc += (double)dy[q * p.C + c] * (double)x[(((long)b * p.H + iy) * p.W + ix) * p.C + c];
    }
  }
  red[part][cl] = acc;
  __syncthreads();
  if (part == 0 && c < p.C) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[q][cl];
    dw[(long)tap * p.C + c] = (float)s;
  }
}

// ---- activations ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float x, int act) {
  if (act == AOT_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if (act == AOT_ACT_RELU6) return (x > 0.f && x < 6.f) ? 1.f : 0.f;
  if (act == AOT_ACT_GELU) {        // d/dx [0.5 x (1 + erf(x / sqrt 2))] = Phi(x) + x phi(x)
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
  }
  if (act == AOT_ACT_SILU) {        // d/dx [x s(x)] = s (1 + x (1 - s))
    const float s = 1.f / (1.f + expf(-x));
    return s * (1.f + x * (1.f - s));
  }
  return 1.f;
}

__global__ void __launch_bounds__(256) act_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = apply_act(x[i], act);
}
__global__ void __launch_bounds__(256) act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ dx, long n, int act) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx[i] = dy[i] * act_grad(x[i], act);
}

// ---- normalisations ---------------------------------------------------------------------------------------------------
// LayerNorm backward, one wave per row (C <= 64 * 32): xhat = (x - mean) rstd (biased variance, eps inside the root, as torch);
// dx = rstd (g - mean(g) - xhat mean(g xhat)),  g = dy gamma
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            float* __restrict__ xhat_out, int M, int C, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (long)row * C;
  const float* gr = dy + (long)row * C;
  double s = 0.0, sq = 0.0;
  for (int c = lane; c < C; c += 64) { const double v = xr[c]; s += v; sq += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); sq += __shfl_xor(sq, o); }
  const double mean = s / C;
  double var = sq / C - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps)), mu = (float)mean;
  double a = 0.0, b = 0.0;
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - mu) * rstd, g = gr[c] * gamma[c];
    a += g;
    b += (double)g * xh;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  const float ma = (float)(a / C), mb = (float)(b / C);
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - mu) * rstd, g = gr[c] * gamma[c];
    dx[(long)row * C + c] = rstd * (g - ma - xh * mb);
    xhat_out[(long)row * C + c] = xh;      // for the parameter gradients (aot_norm_param_grads_f32)
  }
}

// GroupNorm backward over B lanes of [M, C] maps with the forward's statistics [B][G][2] (mean, rstd; fp64): one workgroup
// per (lane, group) computes s1 = sum g, s2 = sum g xhat over its M x C/G elements (g = dy gamma), then
// dx = rstd (g - s1 / n - xhat s2 / n); xhat is written out for the parameter gradients.
__global__ void __launch_bounds__(256) groupnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const double* __restrict__ stats, const float* __restrict__ gamma,
                                                            float* __restrict__ dx, float* __restrict__ xhat_out, int M, int C,
                                                            int G) {
  const int g = blockIdx.x, bl = blockIdx.y, t = threadIdx.x;
  const int cg = C / G;
  const float mean = (float)stats[((long)bl * G + g) * 2], rstd = (float)stats[((long)bl * G + g) * 2 + 1];
  const long base = (long)bl * M * C + (long)g * cg;
  const long n = (long)M * cg;
  __shared__ double red[2][256];
  double s1 = 0.0, s2 = 0.0;
  for (long i = t; i < n; i += 256) {
    const long r = i / cg;
    const int c = (int)(i - r * cg);
    const long off = base + r * C + c;
    const float gg = dy[off] * gamma[g * cg + c], xh = (x[off] - mean) * rstd;
    s1 += gg;
    s2 += (double)gg * xh;
  }
  red[0][t] = s1; red[1][t] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; }
    __syncthreads();
  }
  const float m1 = (float)(red[0][0] / (double)n), m2 = (float)(red[1][0] / (double)n);
  for (long i = t; i < n; i += 256) {
    const long r = i / cg;
    const int c = (int)(i - r * cg);
    const long off = base + r * C + c;
    const float gg = dy[off] * gamma[g * cg + c], xh = (x[off] - mean) * rstd;
    dx[off] = rstd * (gg - m1 - xh * m2);
    xhat_out[off] = xh;
  }
}

// The same backward in two well-filled launches (the batched training step: a decoder map is 1e5 tokens x 16 channels per group,
// which ONE workgroup per (lane, group) walked in ~0.3 ms).  Launch 1: grid (G, B, nchunk), fp64 partials of (s1, s2) per row chunk,
// summed in chunk order by the last arriver of the (lane, group)'s ticket -> m12 [B][G][2] = (s1 / n, s2 / n).  Launch 2:
// elementwise dx and xhat, four channels per thread.
__global__ void __launch_bounds__(256) gn_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const double* __restrict__ stats, const float* __restrict__ gamma,
                                                           double* __restrict__ part, unsigned* __restrict__ ticket, float* __restrict__ m12,
                                                           int M, int C, int G, int nchunk) {
  const int g = blockIdx.x, bl = blockIdx.y, chunk = blockIdx.z, t = threadIdx.x;
  const int cg = C / G;
  const float mean = (float)stats[((long)bl * G + g) * 2], rstd = (float)stats[((long)bl * G + g) * 2 + 1];
  const int per = (M + nchunk - 1) / nchunk;
  const int r0 = chunk * per, r1 = min(M, r0 + per);
  const long base = (long)bl * M * C + (long)g * cg;
  const long n = (long)max(0, r1 - r0) * cg;
  __shared__ double red[2][256];
  __shared__ bool last;
  double s1 = 0.0, s2 = 0.0;
  for (long i = t; i < n; i += 256) {
    const long r = r0 + i / cg;
    const int c = (int)(i % cg);
    const long off = base + r * C + c;
    const float gg = dy[off] * gamma[g * cg + c], xh = (x[off] - mean) * rstd;
    s1 += gg;
    s2 += (double)gg * xh;
  }
  red[0][t] = s1; red[1][t] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; }
    __syncthreads();
  }
  const long slot = (long)bl * G + g;
  if (t == 0) {
    part[(slot * nchunk + chunk) * 2] = red[0][0];
    part[(slot * nchunk + chunk) * 2 + 1] = red[1][0];
    __threadfence();
    last = atomicAdd(ticket + slot, 1u) == (unsigned)nchunk - 1;
  }
  __syncthreads();
  if (!last || t != 0) return;
  __threadfence();
  double a = 0.0, b = 0.0;
  for (int q = 0; q < nchunk; ++q) {
    a += __builtin_nontemporal_load(part + (slot * nchunk + q) * 2);
    b += __builtin_nontemporal_load(part + (slot * nchunk + q) * 2 + 1);
  }
  const double nn = (double)M * cg;
  m12[slot * 2] = (float)(a / nn);
  m12[slot * 2 + 1] = (float)(b / nn);
  ticket[slot] = 0u;
}

__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const double* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ m12, float* __restrict__ dx, float* __restrict__ xhat_out,
                                                           long total4, int M, int C, int G) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
  if (i4 >= total4) return;
  const long e = i4 * 4;
  const long row = e / C;
  const int c = (int)(e - row * C);
  const int bl = (int)(row / M), cg = C / G;
  const float4 xv = *reinterpret_cast<const float4*>(x + e), dv = *reinterpret_cast<const float4*>(dy + e);
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
  float o[4], h[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int g = (c + k) / cg;
    const long slot = (long)bl * G + g;
    const float mean = (float)stats[slot * 2], rstd = (float)stats[slot * 2 + 1];
    const float gg = ds[k] * gamma[c + k], xh = (xs[k] - mean) * rstd;
    o[k] = rstd * (gg - m12[slot * 2] - xh * m12[slot * 2 + 1]);
    h[k] = xh;
  }
  *reinterpret_cast<float4*>(dx + e) = make_float4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<float4*>(xhat_out + e) = make_float4(h[0], h[1], h[2], h[3]);
}

// dgamma[c] = sum_rows dy xhat, dbeta[c] = sum_rows dy over R rows of [R, C]: one workgroup per 8 channels (a few hundred channels
// already make a few dozen workgroups), 32 row slices, fp64 partials summed in fixed order
__global__ void __launch_bounds__(256) norm_param_grads_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, long R,
                                                               int C) {
  const int cl = threadIdx.x & 7, part = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  __shared__ double red[2][32][8];
  double a = 0.0, b = 0.0;
  if (c < C)
    for (long r = part; r < R; r += 32) {
      const float g = dy[r * C + c];
      a += (double)g * xhat[r * C + c];
      b += g;
    }
  red[0][part][cl] = a;
  red[1][part][cl] = b;
  __syncthreads();
  if (part == 0 && c < C) {
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) { sa += red[0][q][cl]; sb += red[1][q][cl]; }
    dgamma[c] = (float)sa;
    dbeta[c] = (float)sb;
  }
}

// The same two column reductions for LONG inputs (the batched step: R = lanes x tokens reaches 1e5 rows): a grid of (32-channel
// group, row chunk); a workgroup is 8 row lanes x 32 channels (every row read is one 128-byte line), fp64 partials per chunk, the
// workgroup that arrives last at a channel group's ticket sums the chunks in order -- deterministic, one launch.
// part [2][nchunk][C] doubles, ticket [ceil(C / 32)] unsigned (zero before the first use; re-armed by the last arriver).
__global__ void __launch_bounds__(256) col_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, long R, int C,
                                                         double* __restrict__ part, unsigned* __restrict__ ticket, int nchunk) {
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, chunk = blockIdx.y;
  const long per = (R + nchunk - 1) / nchunk;
  const long r0 = chunk * per, r1 = min(R, r0 + per);
  __shared__ double red[2][8][32];
  __shared__ bool last;
  double a = 0.0, b = 0.0;
  if (c < C) {
    long r = r0 + rl;
    for (; r + 24 < r1; r += 32) {          // four rows in flight per thread
      float g[4], h[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = dy[(r + 8 * u) * C + c];
      if (xhat) {
#pragma unroll
        for (int u = 0; u < 4; ++u) h[u] = xhat[(r + 8 * u) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a += (double)g[u] * h[u]; b += g[u]; }
    }
    for (; r < r1; r += 8) {
      const float g = dy[r * C + c];
      if (xhat) a += (double)g * xhat[r * C + c];
      b += g;
    }
  }
  red[0][rl][cl] = a;
  red[1][rl][cl] = b;
  __syncthreads();
  if (rl == 0 && c < C) {
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { sa += red[0][q][cl]; sb += red[1][q][cl]; }
    part[(long)chunk * C + c] = sa;
    part[((long)nchunk + chunk) * C + c] = sb;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(ticket + blockIdx.x, 1u) == (unsigned)nchunk - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (rl == 0 && c < C) {
    double sa = 0.0, sb = 0.0;
    for (int q = 0; q < nchunk; ++q) {
      sa += __builtin_nontemporal_load(part + (long)q * C + c);
      sb += __builtin_nontemporal_load(part + ((long)nchunk + q) * C + c);
    }
    if (dgamma) dgamma[c] = (float)sa;
    dbeta[c] = (float)sb;
  }
  if (threadIdx.x == 0) ticket[blockIdx.x] = 0u;
}

// dst [C, Rpad] = src [R, C]^T, columns R..Rpad-1 zero (TR) / dst [Rpad, C] = src rows followed by zero rows (!TR): the operand
// copies of the weight-gradient GEMM (dy^T and x^T with the reduction length padded to the split-K granule) in ONE launch each
// instead of a fill, a strided copy and a concatenation.  32 x 32 tiles through LDS: both sides coalesced.
template <bool TR>
__global__ void __launch_bounds__(256) transpose_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, long R, int C,
                                                            long lds_, long ldd, long Rpad) {
  if (!TR) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= Rpad * C) return;
    const long r = i / C;
    const int c = (int)(i - r * C);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R) v = *reinterpret_cast<const float4*>(src + r * lds_ + c);
    *reinterpret_cast<float4*>(dst + r * ldd + c) = v;
    return;
  }
  __shared__ float tile[32][33];
  const long rb = (long)blockIdx.x * 32;
  const int cb = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long r = rb + ty + 8 * k;
    const int c = cb + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? src[r * lds_ + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = cb + ty + 8 * k;
    const long r = rb + tx;
    if (c < C && r < Rpad) dst[(long)c * ldd + r] = tile[tx][ty + 8 * k];
  }
}

// dst [Rpad, ldd] (columns < Cpad written) = the 2-D view src[r * s_r + c * s_c] (r < R, c < C; ANY strides, zero strides included),
// zeros outside: the operands of the matrix-core GEMMs made from whatever view autograd hands over -- a row-major matrix, a
// transposed view, a broadcast -- with the reduction length padded to the kernels' granule, in ONE launch instead of torch's
// fill + strided copy (+ transpose copy).  32 x 32 tiles; a view whose ROWS are contiguous (s_r == 1: a transposed row-major
// matrix) is read along r and turned through LDS, so that both sides stay coalesced.
__global__ void __launch_bounds__(256) copy2d_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, long R, long C, long s_r,
                                                         long s_c, long Rpad, long Cpad, long ldd) {
  __shared__ float tile[32][33];
  const long rb = (long)blockIdx.x * 32, cb = (long)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (s_r == 1 && s_c != 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long c = cb + ty + 8 * k, r = rb + tx;
      tile[ty + 8 * k][tx] = (r < R && c < C) ? src[r + c * s_c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long r = rb + ty + 8 * k, c = cb + tx;
      if (r < Rpad && c < Cpad) dst[r * ldd + c] = tile[tx][ty + 8 * k];
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long r = rb + ty + 8 * k, c = cb + tx;
    if (r < Rpad && c < Cpad) dst[r * ldd + c] = (r < R && c < C) ? src[r * s_r + c * s_c] : 0.f;
  }
}

// out[r][t] = x[r][idx[t]] for t < Cout (idx int32, a permutation or a selection of the Cin columns): the identity (un)shuffle of
// the logits (trainer.py:457, aot_engine.py:364-367) as a column gather instead of a product with a 0 / 1 matrix
__global__ void __launch_bounds__(256) gather_cols_kernel(const float* __restrict__ x, const int* __restrict__ idx, float* __restrict__ out,
                                                          long R, int Cin, int Cout) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * Cout) return;
  const long r = i / Cout;
  const int t = (int)(i - r * Cout);
  out[i] = x[r * Cin + idx[t]];
}

// ---- softmax over rows ------------------------------------------------------------------------------------------------
// y = softmax(x) per row of length T (entries at -inf give exactly 0); one workgroup per row
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int T) {
  const long row = blockIdx.x;
  const float* xr = x + row * T;
  float* yr = y + row * T;
  const int t = threadIdx.x;
  __shared__ float redf[256];
  __shared__ double redd[256];
  float m = -INFINITY;
  for (int i = t; i < T; i += 256) m = fmaxf(m, xr[i]);
  redf[t] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) redf[t] = fmaxf(redf[t], redf[t + o]);
    __syncthreads();
  }
  m = redf[0];
  double s = 0.0;
  for (int i = t; i < T; i += 256) s += (double)expf(xr[i] - m);
  redd[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) redd[t] += redd[t + o];
    __syncthreads();
  }
  const float inv = (float)(1.0 / redd[0]);
  for (int i = t; i < T; i += 256) yr[i] = expf(xr[i] - m) * inv;
}
// dx = y (dy - sum(dy y))
__global__ void __launch_bounds__(256) softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                               float* __restrict__ dx, int T) {
  const long row = blockIdx.x;
  const float* yr = y + row * T;
  const float* gr = dy + row * T;
  const int t = threadIdx.x;
  __shared__ double redd[256];
  double s = 0.0;
  for (int i = t; i < T; i += 256) s += (double)yr[i] * gr[i];
  redd[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) redd[t] += redd[t + o];
    __syncthreads();
  }
  const float dot = (float)redd[0];
  for (int i = t; i < T; i += 256) dx[row * T + i] = yr[i] * (gr[i] - dot);
}

// ---- bilinear resize, adjoint -----------------------------------------------------------------------------------------
// (same source-index arithmetic as bilinear_kernel in norm_act.hip: ATen's area_pixel_compute_source_index)
__device__ __forceinline__ void bl_coord(int dst, int in_size, int out_size, float scale, int align, int& i0, int& i1,
                                         float& w0, float& w1) {
  if (in_size == out_size) { i0 = i1 = dst; w0 = 1.f; w1 = 0.f; return; }
  float src;
  if (align) src = scale * (float)dst;
  else {
    src = fmaf(scale, (float)dst + 0.5f, -0.5f);
    if (src < 0.f) src = 0.f;
  }
  i0 = min((int)floorf(src), in_size - 1);
  w1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
  w0 = 1.f - w1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
}
inline float bl_scale(int in_size, int out_size, int align) {
  if (align) return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  return (float)in_size / (float)out_size;
}

// dx[(b, iy, ix)][c] = sum over the output pixels whose 2x2 source footprint contains (iy, ix) of weight * dy, in raster order
// of the output: a gather (deterministic).  Candidate output rows / columns come from inverting the source-index map with a
// margin; every candidate is then tested with the exact forward arithmetic.
__global__ void __launch_bounds__(256) bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int IH, int IW,
                                                           int OH, int OW, int C, int align, float sh, float sw) {
  dy += (long)blockIdx.y * OH * OW * C;
  dx += (long)blockIdx.y * IH * IW * C;
  const int nv = C >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)IH * IW * nv) return;
  const int c4 = (int)(idx % nv);
  const int pix = (int)(idx / nv);
  const int iy = pix / IW, ix = pix - iy * IW;
  auto range = [](int i, int in_size, int out_size, float scale, int& lo, int& hi) {
    if (in_size == out_size) { lo = hi = i; return; }
    const float inv = scale > 0.f ? 1.f / scale : (float)out_size;
    lo = max(0, (int)floorf(((float)i - 1.f) * inv) - 2);
    hi = min(out_size - 1, (int)ceilf(((float)i + 1.f) * inv) + 2);
    if (scale <= 0.f) { lo = 0; hi = out_size - 1; }
  };
  int ylo, yhi, xlo, xhi;
  range(iy, IH, OH, sh, ylo, yhi);
  range(ix, IW, OW, sw, xlo, xhi);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int oy = ylo; oy <= yhi; ++oy) {
    int y0, y1;
    float wy0, wy1;
    bl_coord(oy, IH, OH, sh, align, y0, y1, wy0, wy1);
    float wy = 0.f;
    if (y0 == iy) wy += wy0;
    if (y1 == iy) wy += wy1;
    if (wy == 0.f) continue;
    for (int ox = xlo; ox <= xhi; ++ox) {
      int x0, x1;
      float wx0, wx1;
      bl_coord(ox, IW, OW, sw, align, x0, x1, wx0, wx1);
      float wx = 0.f;
      if (x0 == ix) wx += wx0;
      if (x1 == ix) wx += wx1;
      if (wx == 0.f) continue;
      const float4 g = *reinterpret_cast<const float4*>(dy + ((long)oy * OW + ox) * C + c4 * 4);
      const float wgt = wy * wx;
      acc.x = fmaf(wgt, g.x, acc.x); acc.y = fmaf(wgt, g.y, acc.y);
      acc.z = fmaf(wgt, g.z, acc.z); acc.w = fmaf(wgt, g.w, acc.w);
    }
  }
  *reinterpret_cast<float4*>(dx + (long)pix * C + c4 * 4) = acc;
}

// ---- window <-> dense -------------------------------------------------------------------------------------------------
// tokens n = (y, x) of an h x w map, window slots s = (dy + R) * (2R + 1) + (dx + R), key of (n, s) = (y + dy, x + dx).
// gather:  win[g][n][s] = dense[g][n][key(n, s)]  or `fill` where the key lies outside the map
__global__ void __launch_bounds__(256) window_gather_kernel(const float* __restrict__ dense, float* __restrict__ win, int h, int w,
                                                            int R, float fill) {
  const int WS = 2 * R + 1, W2 = WS * WS, N = h * w;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * W2) return;
  const int s = (int)(idx % W2), n = (int)(idx / W2);
  const int y = n / w, x = n - y * w;
  const int ky = y + s / WS - R, kx = x + s % WS - R;
  const long g = blockIdx.y;
  float v = fill;
  if ((unsigned)ky < (unsigned)h && (unsigned)kx < (unsigned)w) v = dense[(g * N + n) * N + (long)ky * w + kx];
  win[(g * N + n) * W2 + s] = v;
}
// scatter (as a gather over the dense side):  dense[g][n][m] = win[g][n][slot(n, m)]  or `fill` where m is outside n's window
__global__ void __launch_bounds__(256) window_scatter_kernel(const float* __restrict__ win, float* __restrict__ dense, int h, int w,
                                                             int R, float fill) {
  const int WS = 2 * R + 1, W2 = WS * WS, N = h * w;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * N) return;
  const int m = (int)(idx % N), n = (int)(idx / N);
  const int y = n / w, x = n - y * w, ky = m / w, kx = m - ky * w;
  const int dy = ky - y, dx = kx - x;
  const long g = blockIdx.y;
  float v = fill;
  if (dy >= -R && dy <= R && dx >= -R && dx <= R) v = win[(g * N + n) * W2 + (dy + R) * WS + (dx + R)];
  dense[(g * N + n) * N + m] = v;
}

}  // namespace

// ===== C ABI ==============================================================================================================
extern "C" int aot_matmul_strided_f32(const float* a, const float* b, const float* bias, float* c, int batch, int M, int N, int K,
                                      long sab, long sam, long sak, long sbb, long sbk, long sbn, long scb, int ldc, float alpha,
                                      int accumulate, void* stream) {
  if (!a || !b || !c || batch <= 0 || M <= 0 || N <= 0 || K <= 0 || ldc < N || batch > 65535) return AOT_ERR_BADARG;
  MatmulParams p;
  p.a = a; p.b = b; p.bias = bias; p.c = c;
  p.sab = sab; p.sam = sam; p.sak = sak; p.sbb = sbb; p.sbk = sbk; p.sbn = sbn; p.scb = scb;
  p.M = M; p.N = N; p.K = K; p.ldc = ldc; p.alpha = alpha; p.accumulate = accumulate;
  if (cdiv(M, 64) > 65535) return AOT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(matmul_strided_kernel, dim3(cdiv(N, 64), cdiv(M, 64), batch), dim3(256), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}

static int fill_col(ColParams& p, int B, int H, int W, int C, int OH, int OW, int KH, int KW, int stride, int pad, int dil) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || OH <= 0 || OW <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 ||
      dil <= 0)
    return AOT_ERR_BADARG;
  p.B = B; p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  return AOT_OK;
}

extern "C" int aot_im2col_f32(const float* x, float* cols, int B, int H, int W, int C, int OH, int OW, int KH, int KW, int stride,
                              int pad, int dil, void* stream) {
  ColParams p;
  if (!x || !cols || fill_col(p, B, H, W, C, OH, OW, KH, KW, stride, pad, dil)) return AOT_ERR_BADARG;
  const long total = (long)B * OH * OW * KH * KW * (C / 4);
  hipLaunchKernelGGL(im2col_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, cols, p);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_col2im_f32(const float* cols, float* dx, int B, int H, int W, int C, int OH, int OW, int KH, int KW, int stride,
                              int pad, int dil, void* stream) {
  ColParams p;
  if (!cols || !dx || fill_col(p, B, H, W, C, OH, OW, KH, KW, stride, pad, dil)) return AOT_ERR_BADARG;
  const long total = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(col2im_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, cols, dx, p);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_dwconv2d_bwd_data_f32(const float* dy, const float* w, float* dx, int B, int H, int W, int C, int OH, int OW,
                                         int KH, int KW, int stride, int pad, int dil, void* stream) {
  ColParams p;
  if (!dy || !w || !dx || fill_col(p, B, H, W, C, OH, OW, KH, KW, stride, pad, dil)) return AOT_ERR_BADARG;
  const long total = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, w, dx, p);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_dwconv2d_bwd_weight_f32(const float* dy, const float* x, float* dw, int B, int H, int W, int C, int OH, int OW,
                                           int KH, int KW, int stride, int pad, int dil, void* stream) {
  ColParams p;
  if (!dy || !x || !dw || fill_col(p, B, H, W, C, OH, OW, KH, KW, stride, pad, dil)) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(dwconv_bwd_weight_kernel, dim3(KH * KW, cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, dy, x, dw, p);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_act_f32(const float* x, float* y, long n, int act, void* stream) {
  if (!x || !y || n <= 0 || act < 0 || act > AOT_ACT_SILU) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(act_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, act);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_act_bwd_f32(const float* x, const float* dy, float* dx, long n, int act, void* stream) {
  if (!x || !dy || !dx || n <= 0 || act < 0 || act > AOT_ACT_SILU) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n, act);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, float* dx, float* xhat, int M, int C,
                                     float eps, void* stream) {
  if (!x || !dy || !gamma || !dx || !xhat || M <= 0 || C <= 0) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, dy, gamma, dx, xhat, M, C, eps);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_groupnorm_bwd_f32(const float* x, const float* dy, const double* stats, const float* gamma, float* dx,
                                     float* xhat, int B, int M, int C, int G, void* stream) {
  if (!x || !dy || !stats || !gamma || !dx || !xhat || B <= 0 || B > 65535 || M <= 0 || C <= 0 || G <= 0 || C % G)
    return AOT_ERR_BADARG;
  hipLaunchKernelGGL(groupnorm_bwd_kernel, dim3(G, B), dim3(256), 0, (hipStream_t)stream, x, dy, stats, gamma, dx, xhat, M, C, G);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_groupnorm_bwd2_f32(const float* x, const float* dy, const double* stats, const float* gamma, float* dx, float* xhat,
                                      double* part, unsigned* ticket, float* m12, int B, int M, int C, int G, int nchunk, void* stream) {
  if (!x || !dy || !stats || !gamma || !dx || !xhat || !part || !ticket || !m12 || B <= 0 || B > 65535 || M <= 0 || C <= 0 || G <= 0 ||
      C % G || (C & 3) || nchunk <= 0 || nchunk > 65535)
    return AOT_ERR_BADARG;
  hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(G, B, nchunk), dim3(256), 0, (hipStream_t)stream, x, dy, stats, gamma, part, ticket, m12, M, C,
                     G, nchunk);
  const long total4 = (long)B * M * C / 4;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, stats, gamma, m12, dx, xhat,
                     total4, M, C, G);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_norm_param_grads_f32(const float* dy, const float* xhat, float* dgamma, float* dbeta, long R, int C,
                                        void* stream) {
  if (!dy || !xhat || !dgamma || !dbeta || R <= 0 || C <= 0) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(norm_param_grads_kernel, dim3(cdiv(C, 8)), dim3(256), 0, (hipStream_t)stream, dy, xhat, dgamma, dbeta, R, C);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_col_reduce_f32(const float* dy, const float* xhat, float* dgamma, float* dbeta, long R, int C, double* part,
                                  unsigned* ticket, int nchunk, void* stream) {
  if (!dy || !dbeta || !part || !ticket || R <= 0 || C <= 0 || nchunk <= 0 || nchunk > 65535 || (xhat && !dgamma)) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(col_reduce_kernel, dim3(cdiv(C, 32), nchunk), dim3(256), 0, (hipStream_t)stream, dy, xhat, dgamma, dbeta, R, C, part,
                     ticket, nchunk);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_transpose_pad_f32(const float* src, float* dst, long R, int C, long lds, long ldd, long Rpad, int transpose,
                                     void* stream) {
  if (!src || !dst || R <= 0 || C <= 0 || Rpad < R || lds < C) return AOT_ERR_BADARG;
  if (transpose) {
    if (ldd < Rpad || cdiv(C, 32) > 65535) return AOT_ERR_BADARG;
    hipLaunchKernelGGL((transpose_pad_kernel<true>), dim3(cdiv(Rpad, 32), cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, src, dst, R, C, lds,
                       ldd, Rpad);
  } else {
    if ((C & 3) || (lds & 3) || (ldd & 3) || ldd < C || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return AOT_ERR_BADARG;
    hipLaunchKernelGGL((transpose_pad_kernel<false>), dim3(cdiv(Rpad * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, R, C, lds,
                       ldd, Rpad);
  }
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_copy2d_pad_f32(const float* src, float* dst, long R, long C, long s_r, long s_c, long Rpad, long Cpad, long ldd,
                                  void* stream) {
  if (!src || !dst || R <= 0 || C <= 0 || s_r < 0 || s_c < 0 || Rpad < R || Cpad < C || ldd < Cpad) return AOT_ERR_BADARG;
  if (cdiv(Cpad, 32) > 65535 || cdiv(Rpad, 32) > 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(copy2d_pad_kernel, dim3((unsigned)cdiv(Rpad, 32), (unsigned)cdiv(Cpad, 32)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     R, C, s_r, s_c, Rpad, Cpad, ldd);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_gather_cols_f32(const float* x, const int* idx, float* out, long R, int Cin, int Cout, void* stream) {
  if (!x || !idx || !out || R <= 0 || Cin <= 0 || Cout <= 0) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(R * Cout, 256)), dim3(256), 0, (hipStream_t)stream, x, idx, out, R, Cin, Cout);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_softmax_rows_f32(const float* x, float* y, long rows, int T, void* stream) {
  if (!x || !y || rows <= 0 || T <= 0 || rows > 0x7fffffffL) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, y, T);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, long rows, int T, void* stream) {
  if (!y || !dy || !dx || rows <= 0 || T <= 0 || rows > 0x7fffffffL) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, y, dy, dx, T);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_bilinear_bwd_nhwc_f32(const float* dy, float* dx, int B, int IH, int IW, int OH, int OW, int C,
                                         int align_corners, void* stream) {
  if (!dy || !dx || B <= 0 || B > 65535 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0 || (C & 3)) return AOT_ERR_BADARG;
  const long total = (long)IH * IW * (C / 4);
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(cdiv(total, 256), B), dim3(256), 0, (hipStream_t)stream, dy, dx, IH, IW, OH, OW, C,
                     align_corners, bl_scale(IH, OH, align_corners), bl_scale(IW, OW, align_corners));
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_window_gather_f32(const float* dense, float* win, int G, int h, int w, int max_dis, float fill, void* stream) {
  if (!dense || !win || G <= 0 || G > 65535 || h <= 0 || w <= 0 || max_dis < 0) return AOT_ERR_BADARG;
  const long total = (long)h * w * (2 * max_dis + 1) * (2 * max_dis + 1);
  hipLaunchKernelGGL(window_gather_kernel, dim3(cdiv(total, 256), G), dim3(256), 0, (hipStream_t)stream, dense, win, h, w, max_dis,
                     fill);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_window_scatter_f32(const float* win, float* dense, int G, int h, int w, int max_dis, float fill, void* stream) {
  if (!win || !dense || G <= 0 || G > 65535 || h <= 0 || w <= 0 || max_dis < 0) return AOT_ERR_BADARG;
  const long total = (long)h * w * h * w;
  hipLaunchKernelGGL(window_scatter_kernel, dim3(cdiv(total, 256), G), dim3(256), 0, (hipStream_t)stream, win, dense, h, w, max_dis,
                     fill);
  AOT_LAUNCH_CHECK();
}
