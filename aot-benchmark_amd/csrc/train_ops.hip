// Training-side stages (first slice): the two segmentation losses of the reference with their gradients, the
// AdamW step, the EMA update and the gradient-norm reduction.  Streaming kernels (HBM-bound: bytes in + out), fp32 data with
// fp64 accumulation wherever a sum runs over pixels, fixed summation order (deterministic).
//
//   networks/layers/loss.py:137-188  CrossEntropyLoss(top_k_percent_pixels, hard_example_mining_step)
//   networks/layers/loss.py:119-137  SoftJaccordLoss  = tversky_loss(alpha = beta = 1), loss.py:29-55, on flatten_probas(:58-72)
//   networks/managers/trainer.py:116-118  torch.optim.AdamW ; trainer.py:501-503 clip_grad_norm_ ; utils/ema.py:52-66
//
// Layouts: logits [B, C, P] planar (P = H*W, channel stride P, sample stride C*P -- the reference's [B,C,H,W]); labels [B, P]
// fp32 class ids (what the reference's masks are), 255 = ignore.
#include "common.h"
#include <cmath>

namespace {

constexpr int MAXC = 16;
constexpr float IGNORE = 255.f;

__device__ __forceinline__ unsigned ord_u32_t(float f) {      // order-preserving map float -> uint
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ---- cross entropy ----------------------------------------------------------------------------------------------------
// per-pixel loss = logsumexp(z) - z[label]; 0 for ignored pixels (F.cross_entropy(reduction='none', ignore_index=255))
__global__ void __launch_bounds__(256) ce_pixel_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                       float* __restrict__ loss, int C, long P) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const int b = blockIdx.y;
  const float* z = logits + (long)b * C * P + i;
  const float lab = labels[(long)b * P + i];
  float v[MAXC], mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) { v[c] = z[(long)c * P]; mx = fmaxf(mx, v[c]); }
  float s = 0.f, zl = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) { s += expf(v[c] - mx); if ((float)c == lab) zl = v[c]; }
  const bool valid = lab != IGNORE && lab >= 0.f && lab < (float)C;
  loss[(long)b * P + i] = valid ? (mx + logf(s)) - zl : 0.f;
}

// Hard-example mining: mean of the k largest per-pixel losses of one sample (torch.topk + mean, loss.py:176-180).  One
// 1024-thread block per sample: 4-pass 8-bit radix select of the k-th largest, then the sum of everything above it plus the
// wanted share of the ties (equal values: which ones are taken does not change the sum).  For the backward pass it also returns
// the order key of the k-th largest (thr_out[b]) and the share of the pixels tied with it that entered the mean
// (thr_out[B + b], float bits: need / number of ties -- torch.topk takes `need` of them, which ones is unspecified; spreading
// their weight over all of them is the symmetric sub-gradient, and equals torch's whenever there is no tie).
__global__ void __launch_bounds__(1024) topk_mean_kernel(const float* __restrict__ vals, long P, long k, float* __restrict__ mean_out,
                                                         unsigned* __restrict__ thr_out) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* s = vals + (long)b * P;
  __shared__ unsigned hist[256];
  __shared__ unsigned long long sh_need;
  __shared__ unsigned sh_prefix;
  __shared__ double red[1024];
  unsigned prefix = 0, mask = 0;
  unsigned long long need = (unsigned long long)k;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (long t = tid; t < P; t += 1024) {
      const unsigned u = ord_u32_t(s[t]);
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned long long cum = 0;
      int bb = 255;
      for (; bb > 0; --bb) {
        if (cum + hist[bb] >= need) break;
        cum += hist[bb];
      }
      sh_prefix = prefix | ((unsigned)bb << shift);
      sh_need = need - cum;
    }
    __syncthreads();
    prefix = sh_prefix;
    need = sh_need;
    mask |= 0xFFu << shift;
    __syncthreads();
  }
  double acc = 0.0;
  float vthr = 0.f;
  unsigned nt = 0;
  for (long t = tid; t < P; t += 1024) {
    const float f = s[t];
    const unsigned u = ord_u32_t(f);
    if (u > prefix) acc += (double)f;
    if (u == prefix) { vthr = f; ++nt; }
  }
  __shared__ unsigned sh_ties;
  if (tid == 0) sh_ties = 0;
  __syncthreads();
  if (nt) atomicAdd(&sh_ties, nt);      // integer sum: order-independent
  red[tid] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  // the value at the threshold (any thread that saw it holds the same bits)
  __shared__ float sh_v;
  if (tid == 0) sh_v = 0.f;
  __syncthreads();
  if (vthr != 0.f) sh_v = vthr;     // benign race: identical values
  __syncthreads();
  if (tid == 0) {
    mean_out[b] = (float)((red[0] + (double)need * (double)sh_v) / (double)k);
    thr_out[b] = prefix;
    thr_out[gridDim.x + b] = __float_as_uint((float)((double)need / (double)(sh_ties ? sh_ties : 1u)));
  }
}

// plain mean over the valid pixels of the whole batch (nn.CrossEntropyLoss(reduction='mean', ignore_index=255) on one sample:
// the reference calls it per sample, loss.py:160-163): per-sample sum / count in fp64
__global__ void __launch_bounds__(1024) valid_mean_kernel(const float* __restrict__ vals, const float* __restrict__ labels, long P,
                                                          int C, float* __restrict__ mean_out, float* __restrict__ cnt_out) {
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ double rs[1024], rc[1024];
  double acc = 0.0, cnt = 0.0;
  for (long t = tid; t < P; t += 1024) {
    const float lab = labels[(long)b * P + t];
    if (lab != IGNORE && lab >= 0.f && lab < (float)C) { acc += (double)vals[(long)b * P + t]; cnt += 1.0; }
  }
  rs[tid] = acc; rc[tid] = cnt;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) { rs[tid] += rs[tid + o]; rc[tid] += rc[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) { mean_out[b] = (float)(rs[0] / rc[0]); cnt_out[b] = (float)rc[0]; }
}

// d(loss_b)/d(logits): gscale[b] * (softmax - onehot) on the pixels that entered the mean (loss >= the k-th largest when
// thr != nullptr, every valid pixel otherwise), 0 elsewhere and on ignored pixels
__global__ void __launch_bounds__(256) ce_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                     const float* __restrict__ loss, const unsigned* __restrict__ thr,
                                                     const float* __restrict__ gscale, float* __restrict__ grad, int C, long P) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const int b = blockIdx.y;
  const float* z = logits + (long)b * C * P + i;
  float* g = grad + (long)b * C * P + i;
  const float lab = labels[(long)b * P + i];
  const bool valid = lab != IGNORE && lab >= 0.f && lab < (float)C;
  const unsigned key = thr ? ord_u32_t(loss[(long)b * P + i]) : 0u;
  const bool take = valid && (thr == nullptr || key >= thr[b]);
  if (!take) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) g[(long)c * P] = 0.f;
    return;
  }
  float v[MAXC], mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) { v[c] = z[(long)c * P]; mx = fmaxf(mx, v[c]); }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) { v[c] = expf(v[c] - mx); s += v[c]; }
  // a pixel tied with the k-th largest loss carries the share of the ties the forward mean took
  const float gs = gscale[b] * ((thr && key == thr[b]) ? __uint_as_float(thr[gridDim.y + b]) : 1.f);
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) g[(long)c * P] = gs * (v[c] / s - ((float)c == lab ? 1.f : 0.f));
}

// ---- soft Jaccard -----------------------------------------------------------------------------------------------------
// pass 1: per (sample, chunk of pixels, class): I = sum p*g, Sp = sum p, Sg = sum g over the valid pixels, fp64
__global__ void __launch_bounds__(256) jaccard_partial_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                              double* __restrict__ part, int C, long P, int nchunk) {
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const long per = (P + nchunk - 1) / nchunk, p0 = (long)chunk * per, p1 = min(P, p0 + per);
  double aI[MAXC], aP[MAXC], aG[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) aI[c] = aP[c] = aG[c] = 0.0;
  for (long i = p0 + tid; i < p1; i += 256) {
    const float lab = labels[(long)b * P + i];
    if (lab == IGNORE) continue;
    const float* z = logits + (long)b * C * P + i;
    float v[MAXC], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { v[c] = z[(long)c * P]; mx = fmaxf(mx, v[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { v[c] = expf(v[c] - mx); s += v[c]; }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        const float pr = v[c] / s;
        const bool fg = (float)c == lab;
        aP[c] += (double)pr;
        if (fg) { aI[c] += (double)pr; aG[c] += 1.0; }
      }
  }
  __shared__ double red[256];
  double* dst = part + (((long)b * nchunk + chunk) * MAXC) * 3;
  for (int c = 0; c < C; ++c)
    for (int q = 0; q < 3; ++q) {
      red[tid] = q == 0 ? aI[c] : q == 1 ? aP[c] : aG[c];
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
      }
      if (tid == 0) dst[c * 3 + q] = red[0];
      __syncthreads();
    }
}

// pass 2: chunks summed in order; loss_b = mean over the classes present in the sample of 1 - I / (I + FP + FN + eps) with
// FP = Sp - I, FN = Sg - I (tversky alpha = beta = 1); sums [B][C][3] kept for the backward pass, present-class count too
__global__ void __launch_bounds__(64) jaccard_final_kernel(const double* __restrict__ part, double* __restrict__ sums,
                                                           float* __restrict__ loss, int C, int nchunk, float eps) {
  const int b = blockIdx.x, c = threadIdx.x;
  __shared__ double lc[MAXC];
  __shared__ int present[MAXC];
  if (c < C) {
    double s[3] = {0.0, 0.0, 0.0};
    for (int ch = 0; ch < nchunk; ++ch)
      for (int q = 0; q < 3; ++q) s[q] += part[(((long)b * nchunk + ch) * MAXC + c) * 3 + q];
    for (int q = 0; q < 3; ++q) sums[((long)b * MAXC + c) * 3 + q] = s[q];
    present[c] = s[2] > 0.0;
    // the reference evaluates in fp32: numerator I, denominator I + (Sp - I) + (Sg - I) + eps
    const float I = (float)s[0], FP = (float)(s[1] - s[0]), FN = (float)(s[2] - s[0]);
    lc[c] = 1.0 - (double)(I / (I + FP + FN + eps));
  }
  __syncthreads();
  if (c == 0) {
    double acc = 0.0;
    int n = 0;
    for (int cc = 0; cc < C; ++cc)
      if (present[cc]) { acc += lc[cc]; ++n; }
    loss[b] = n ? (float)(acc / n) : 0.f;
  }
}

// backward: dL/dp_c = -(w / n_present) * (g*D - I*(1 - g)) / D^2 with D = Sp + Sg - I + eps, then through the softmax
__global__ void __launch_bounds__(256) jaccard_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                          const double* __restrict__ sums, const float* __restrict__ gout,
                                                          float* __restrict__ grad, int C, long P, float eps) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  __shared__ float sI[MAXC], sD[MAXC], sW[MAXC];
  if (threadIdx.x < MAXC) {
    const int c = threadIdx.x;
    float w = 0.f, I = 0.f, D = 1.f;
    if (c < C) {
      int n = 0;
      for (int cc = 0; cc < C; ++cc) n += sums[((long)b * MAXC + cc) * 3 + 2] > 0.0;
      const double* s = sums + ((long)b * MAXC + c) * 3;
      I = (float)s[0];
      D = I + (float)(s[1] - s[0]) + (float)(s[2] - s[0]) + eps;
      w = (s[2] > 0.0 && n > 0) ? gout[b] / (float)n : 0.f;
    }
    sI[c] = I; sD[c] = D; sW[c] = w;
  }
  __syncthreads();
  if (i >= P) return;
  const float* z = logits + (long)b * C * P + i;
  float* g = grad + (long)b * C * P + i;
  const float lab = labels[(long)b * P + i];
  if (lab == IGNORE) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) g[(long)c * P] = 0.f;
    return;
  }
  float v[MAXC], mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) { v[c] = z[(long)c * P]; mx = fmaxf(mx, v[c]); }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) { v[c] = expf(v[c] - mx); s += v[c]; }
  float dp[MAXC], dot = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) {
      v[c] /= s;
      const float gg = (float)c == lab ? 1.f : 0.f;
      dp[c] = -sW[c] * (gg * sD[c] - sI[c] * (1.f - gg)) / (sD[c] * sD[c]);
      dot += v[c] * dp[c];
    }
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) g[(long)c * P] = v[c] * (dp[c] - dot);
}

// ---- optimiser / EMA ----------------------------------------------------------------------------------------------------
// torch.optim.AdamW (decoupled weight decay), one tensor: p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).  gscale folds the gradient-clipping factor (clip_grad_norm_) in.
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float wd, float b1, float b2, float eps,
                                                    float bc1, float bc2_sqrt, float gscale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  float pi = p[i] * (1.f - lr * wd);
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  pi -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  p[i] = pi;
}

// utils/ema.py:63-66: shadow -= (1 - decay) * (shadow - param)
__global__ void __launch_bounds__(256) ema_kernel(float* __restrict__ s, const float* __restrict__ p, long n, float omd) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) s[i] -= omd * (s[i] - p[i]);
}

// sum of squares of one tensor, accumulated into out[0] in fp64 by ONE workgroup (fixed order: deterministic; gradients of a
// whole model are a few hundred tensors of <= a few MB)
__global__ void __launch_bounds__(1024) sumsq_kernel(const float* __restrict__ x, long n, double* __restrict__ out) {
  __shared__ double red[1024];
  double acc = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) acc += (double)x[i] * (double)x[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] += red[0];
}

// ---- the optimiser over FLAT state (utils/flat_state.py): every trainable tensor is a range of one parameter / gradient /
// moment buffer, so one step is three launches whatever the number of tensors ----------------------------------------------
// sum of squares of the whole gradient buffer: per-block partials in fp64, summed in block order by the block that arrives
// last (ticket) -> deterministic, one launch.  part [gridDim.x] doubles, ticket one unsigned (zero before the first launch; the
// last arriver re-arms it).
__global__ void __launch_bounds__(256) sumsq_flat_kernel(const float* __restrict__ x, long n, double* __restrict__ part,
                                                         unsigned* __restrict__ ticket, double* __restrict__ out) {
  __shared__ double red[256];
  __shared__ bool last;
  double acc = 0.0;
  const long stride = (long)gridDim.x * 1024;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 t = *reinterpret_cast<const float4*>(x + i);
      acc += (double)t.x * t.x + (double)t.y * t.y + (double)t.z * t.z + (double)t.w * t.w;
    } else {
      for (long j = i; j < n; ++j) acc += (double)x[j] * x[j];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[blockIdx.x] = red[0];
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double t = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += __builtin_nontemporal_load(part + i);     // fixed assignment of blocks to threads
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = red[0];
    *ticket = 0u;
  }
}

// AdamW over the flat buffers.  seg_off [nseg + 1] element offsets of the tensors (ascending), hyp [nseg][4] = lr, weight decay,
// bias correction 1, sqrt(bias correction 2) of each tensor at its own step count; lr < 0 marks a tensor that received no
// gradient this step (torch.optim.AdamW skips `p.grad is None`: no decay, no moment update).  The clip factor is computed here
// from the device-resident sum of squares (clip_grad_norm_: max_norm / (norm + 1e-6), capped at 1; max_norm <= 0: no clipping),
// so the step needs no host synchronisation.
__global__ void __launch_bounds__(256) adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long n, const long* __restrict__ seg_off,
                                                         const float* __restrict__ hyp, int nseg, float b1, float b2, float eps,
                                                         const double* __restrict__ sumsq, float max_norm) {
  const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  float gscale = 1.f;
  if (max_norm > 0.f && sumsq) gscale = fminf(1.f, max_norm / ((float)sqrt(sumsq[0]) + 1e-6f));
  // segment of the first element: binary search (the table is a few KB: L1 / L2 resident)
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg_off[mid] <= i0) lo = mid; else hi = mid - 1;
  }
  int sg = lo;
  long end = seg_off[sg + 1];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long i = i0 + e;
    if (i >= n) return;
    while (i >= end) end = seg_off[++sg + 1];
    const float lr = hyp[4 * sg];
    if (lr < 0.f) continue;
    const float wd = hyp[4 * sg + 1], bc1 = hyp[4 * sg + 2], bc2_sqrt = hyp[4 * sg + 3];
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    p[i] = pi;
  }
}

}  // namespace

extern "C" int aot_sumsq_flat_f64(const float* x, long n, double* part, int nblk, unsigned* ticket, double* out, void* stream) {
  if (!x || !part || !ticket || !out || n <= 0 || nblk <= 0 || nblk > 65535 || ((uintptr_t)x & 15)) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(sumsq_flat_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, n, part, ticket, out);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_adamw_flat_f32(float* p, const float* g, float* m, float* v, long n, const long* seg_off, const float* hyp, int nseg,
                                  float beta1, float beta2, float eps, const double* sumsq, float max_norm, void* stream) {
  if (!p || !g || !m || !v || !seg_off || !hyp || n <= 0 || nseg <= 0) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(adamw_flat_kernel, dim3(cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, seg_off, hyp, nseg, beta1,
                     beta2, eps, sumsq, max_norm);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_ce_loss_f32(const float* logits, const float* labels, float* loss_px, float* loss, unsigned* thr, float* cnt,
                               int B, int C, long P, long top_k, void* stream) {
  if (!logits || !labels || !loss_px || !loss || B <= 0 || C <= 1 || P <= 0) return AOT_ERR_BADARG;
  if (C > MAXC) return AOT_ERR_UNSUPPORTED;
  if (top_k > P || (top_k > 0 && !thr) || (top_k <= 0 && !cnt)) return AOT_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_pixel_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, s, logits, labels, loss_px, C, P);
  if (top_k > 0)
    hipLaunchKernelGGL(topk_mean_kernel, dim3(B), dim3(1024), 0, s, loss_px, P, top_k, loss, thr);
  else
    hipLaunchKernelGGL(valid_mean_kernel, dim3(B), dim3(1024), 0, s, loss_px, labels, P, C, loss, cnt);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_ce_loss_bwd_f32(const float* logits, const float* labels, const float* loss_px, const unsigned* thr,
                                   const float* gscale, float* grad, int B, int C, long P, void* stream) {
  if (!logits || !labels || !loss_px || !gscale || !grad || B <= 0 || C <= 1 || P <= 0) return AOT_ERR_BADARG;
  if (C > MAXC) return AOT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, (hipStream_t)stream, logits, labels, loss_px, thr, gscale,
                     grad, C, P);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_soft_jaccard_f32(const float* logits, const float* labels, double* part, double* sums, float* loss, int B,
                                    int C, long P, int nchunk, float eps, void* stream) {
  if (!logits || !labels || !part || !sums || !loss || B <= 0 || C <= 1 || P <= 0 || nchunk <= 0) return AOT_ERR_BADARG;
  if (C > MAXC) return AOT_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(jaccard_partial_kernel, dim3(nchunk, B), dim3(256), 0, s, logits, labels, part, C, P, nchunk);
  hipLaunchKernelGGL(jaccard_final_kernel, dim3(B), dim3(64), 0, s, part, sums, loss, C, nchunk, eps);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_soft_jaccard_bwd_f32(const float* logits, const float* labels, const double* sums, const float* gout,
                                        float* grad, int B, int C, long P, float eps, void* stream) {
  if (!logits || !labels || !sums || !gout || !grad || B <= 0 || C <= 1 || P <= 0) return AOT_ERR_BADARG;
  if (C > MAXC) return AOT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(jaccard_bwd_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, (hipStream_t)stream, logits, labels, sums, gout,
                     grad, C, P, eps);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_adamw_step_f32(float* p, const float* g, float* m, float* v, long n, float lr, float weight_decay, float beta1,
                                  float beta2, float eps, int step, float gscale, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) return AOT_ERR_BADARG;
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));          // in double on the host, like torch's python scalars
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, weight_decay, beta1,
                     beta2, eps, bc1, bc2_sqrt, gscale);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_ema_update_f32(float* shadow, const float* param, long n, float one_minus_decay, void* stream) {
  if (!shadow || !param || n <= 0) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(ema_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, shadow, param, n, one_minus_decay);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_sumsq_accum_f64(const float* x, long n, double* out, void* stream) {
  if (!x || !out || n <= 0) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out);
  AOT_LAUNCH_CHECK();
}
