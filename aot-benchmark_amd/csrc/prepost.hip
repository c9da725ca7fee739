// The steps either side of the engine in the reference's evaluator, as device kernels:
//   aot_preprocess_f32    MultiRestrictSize's cubic resize (+ flip) and MultiToTensor's normalisation
//                         (dataloaders/video_transforms.py:594-715)  -> engine input [1,3,OH,OW]
//   aot_fuse_probs_f32    per-augmentation softmax, un-flip, mean over augmentations, argmax, new-object merge
//                         (networks/managers/evaluator.py:325-372)
//   aot_label_resize_f32  flip + F.interpolate(mode="nearest") of a label map to an engine's input size (:383-386,405-408)
// All are HBM streamers: one thread per output pixel, coalesced along x.
#include "common.h"

// OpenCV INTER_CUBIC (imgproc/resize.cpp, interpolateCubic, A = -0.75): weights of taps -1, 0, +1, +2 at fraction x.
__device__ __forceinline__ void cubic_coeffs(float x, float (&c)[4]) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
  c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  c[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

struct PreParams {
  const void* src;
  float* dst;
  int u8, H, W, lds, OH, OW, flip;
  double mean[3], stdv[3];
};

__device__ __forceinline__ float px(const PreParams& p, int y, int x, int c) {
  const long i = (long)y * p.lds + (long)x * 3 + c;
  return p.u8 ? (float)reinterpret_cast<const unsigned char*>(p.src)[i] : reinterpret_cast<const float*>(p.src)[i];
}

__global__ void __launch_bounds__(256) preprocess_kernel(const PreParams p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= p.OW) return;
  float v[3];
  if (p.OH == p.H && p.OW == p.W) {     // MultiRestrictSize keeps the sample untouched when the size already fits (:655-656)
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = px(p, y, x, c);
  } else {
    // cv2.resize: fx = (dx + 0.5) * scale - 0.5 in double -> float, sx = floor(fx), taps clamped to the image (replicate)
    const double sy_d = (double)p.H / p.OH, sx_d = (double)p.W / p.OW;
    float fy = (float)((y + 0.5) * sy_d - 0.5), fx = (float)((x + 0.5) * sx_d - 0.5);
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    fy -= iy; fx -= ix;
    float cy[4], cx[4];
    cubic_coeffs(fy, cy);
    cubic_coeffs(fx, cx);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yy = min(max(iy - 1 + j, 0), p.H - 1);
      float r[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {      // horizontal pass first, as cv2 does
        const int xx = min(max(ix - 1 + i, 0), p.W - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] += px(p, yy, xx, c) * cx[i];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] += r[c] * cy[j];
    }
  }
  // MultiToTensor (:703-711): tmp = tmp / 255. (float32); tmp -= mean; tmp /= std  (numpy computes the in-place ops with
  // the float64 tuple in double and rounds back to float32)
  const int ox = p.flip ? p.OW - 1 - x : x;       // flipped sample = resized sample reversed along x (:669-680)
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float t = v[c] / 255.f;
    t = (float)((double)t - p.mean[c]);
    t = (float)((double)t / p.stdv[c]);
    p.dst[((long)c * p.OH + y) * p.OW + ox] = t;
  }
}

extern "C" int aot_preprocess_f32(const void* src, int src_is_u8, int H, int W, int ld_src, float* dst, int OH, int OW,
                                  int flip, const double* mean3, const double* std3, void* stream) {
  if (!src || !dst || !mean3 || !std3 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || ld_src < 3 * W) return AOT_ERR_BADARG;
  PreParams p;
  p.src = src; p.dst = dst; p.u8 = src_is_u8; p.H = H; p.W = W; p.lds = ld_src; p.OH = OH; p.OW = OW; p.flip = flip;
  for (int c = 0; c < 3; ++c) {
    if (!(std3[c] != 0.0)) return AOT_ERR_BADARG;
    p.mean[c] = mean3[c]; p.stdv[c] = std3[c];
  }
  hipLaunchKernelGGL(preprocess_kernel, dim3(cdiv(OW, 256), OH), dim3(256), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}

struct FuseParams {
  const float* logits;      // [A][nc][HW]
  const float* new_label;   // optional [HW]: non-zero pixels override every label (new objects, evaluator.py:362-369)
  float* fused_label;       // [HW]
  float* aug_labels;        // optional [A][HW]: argmax of every augmentation's own probabilities (un-flipped frame)
  float* fused_prob;        // optional [nc][HW]: mean probability
  int A, nc, OH, OW, flipmask;
};

// MAXC = register-array bound on the channel count (1 + 10 objects per group; datasets/Demo: 44 objects -> 51 channels)
template <int AOT_FUSE_MAXC>
__global__ void __launch_bounds__(256) fuse_probs_kernel(const FuseParams p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= p.OW) return;
  const long HW = (long)p.OH * p.OW, pix = (long)y * p.OW + x;
  float acc[AOT_FUSE_MAXC];
#pragma unroll
  for (int c = 0; c < AOT_FUSE_MAXC; ++c) acc[c] = 0.f;
  const float nl = p.new_label ? p.new_label[pix] : 0.f;
  for (int a = 0; a < p.A; ++a) {
    const int sx = ((p.flipmask >> a) & 1) ? p.OW - 1 - x : x;     // flip_tensor(pred_logit, 3) (:329-330)
    const float* lg = p.logits + (long)a * p.nc * HW + (long)y * p.OW + sx;
    float v[AOT_FUSE_MAXC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < AOT_FUSE_MAXC; ++c)
      if (c < p.nc) { v[c] = lg[(long)c * HW]; m = fmaxf(m, v[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < AOT_FUSE_MAXC; ++c)
      if (c < p.nc) { v[c] = expf(v[c] - m); s += v[c]; }
    int best = 0;
    float bv = -1.f;
#pragma unroll
    for (int c = 0; c < AOT_FUSE_MAXC; ++c)
      if (c < p.nc) {
        const float pr = v[c] / s;              // torch.softmax(pred_logit, dim=1) (:332)
        acc[c] += pr;
        if (pr > bv) { bv = pr; best = c; }      // first maximum, as torch.argmax
      }
    if (p.aug_labels) p.aug_labels[(long)a * HW + pix] = (nl != 0.f) ? nl : (float)best;
  }
  int best = 0;
  float bv = -1.f;
#pragma unroll
  for (int c = 0; c < AOT_FUSE_MAXC; ++c)
    if (c < p.nc) {
      const float pr = acc[c] / (float)p.A;     // torch.mean(cat_all_preds, dim=0) (:349-352)
      if (p.fused_prob) p.fused_prob[(long)c * HW + pix] = pr;
      if (pr > bv) { bv = pr; best = c; }
    }
  p.fused_label[pix] = (nl != 0.f) ? nl : (float)best;       // pred_label * keep + new_obj_label * (1 - keep) (:362-369)
}

extern "C" int aot_fuse_probs_f32(const float* logits, const float* new_label, float* fused_label, float* aug_labels,
                                  float* fused_prob, int A, int nc, int OH, int OW, int flipmask, void* stream) {
  if (!logits || !fused_label || A <= 0 || A > 30 || nc <= 0 || nc > 64 || OH <= 0 || OW <= 0) return AOT_ERR_BADARG;
  FuseParams p;
  p.logits = logits; p.new_label = new_label; p.fused_label = fused_label; p.aug_labels = aug_labels;
  p.fused_prob = fused_prob; p.A = A; p.nc = nc; p.OH = OH; p.OW = OW; p.flipmask = flipmask;
  const dim3 grid(cdiv(OW, 256), OH);
  if (nc <= 16) hipLaunchKernelGGL(fuse_probs_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else if (nc <= 32) hipLaunchKernelGGL(fuse_probs_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(fuse_probs_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}

__global__ void __launch_bounds__(256) label_resize_kernel(const float* src, float* dst, int H, int W, int OH, int OW,
                                                           int flip) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= OW) return;
  // torch upsample_nearest2d ("nearest", legacy): src = min(floor(dst * (float)in / out), in - 1)
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  const int sy = min((int)floorf(y * sh), H - 1);
  int sx = min((int)floorf(x * sw), W - 1);
  if (flip) sx = W - 1 - sx;                    // flip_tensor(label, 3) precedes the interpolation (:375-381)
  dst[(long)y * OW + x] = src[(long)sy * W + sx];
}

extern "C" int aot_label_resize_f32(const float* src, float* dst, int H, int W, int OH, int OW, int flip, void* stream) {
  if (!src || !dst || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(label_resize_kernel, dim3(cdiv(OW, 256), OH), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, OH, OW, flip);
  AOT_LAUNCH_CHECK();
}
