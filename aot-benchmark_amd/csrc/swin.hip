// Swin-B encoder pieces (reference networks/encoders/swin/swin_transformer.py).
//
// swin_window_attn_kernel: W-MSA / SW-MSA of one 7x7 window and one head per 64-lane wave.  The reference pads
// the LayerNorm output to multiples of 7, rolls it by -shift, partitions windows, runs attention with the
// relative-position bias and the -100 region mask, then reverses partition, roll and crop (:262-318).  All of
// that is index arithmetic here: lane i (< 49) is token (i/7, i%7) of window (WY, WX) of the SHIFTED padded map,
// which is token ((WY*7 + i/7 + shift) % Hp, (WX*7 + i%7 + shift) % Wp) of the un-shifted map; padded tokens
// (outside H x W) carry qkv = bias, exactly what the reference's Linear gives for the zero rows it pads in.
// K and V of the window sit in LDS and are read as broadcasts; 2 x 49 x 32 FMAs per lane.
#include "common.h"

struct SwinAttnParams {
  const float* qkv;    // [H*W, ld] columns: q | k | v, each C = nH*32 wide
  const float* bias;   // [3C] qkv bias (value of padded tokens)
  const float* table;  // [169, nH] relative_position_bias_table
  float* out;          // [H*W, ldo]
  int H, W, C, nH, ld, ldo, shift, Hp, Wp;
  float scale;
  long img_rows;       // rows between the maps of consecutive images of the batch (blockIdx.z): H*W
};

__global__ void __launch_bounds__(64) swin_window_attn_kernel(const SwinAttnParams p) {
  constexpr int WS = 7, NT = 49, D = 32;
  __shared__ __attribute__((aligned(16))) float Ks[NT][D + 4];
  __shared__ __attribute__((aligned(16))) float Vs[NT][D + 4];
  __shared__ float tbl[176];
  __shared__ int lab[NT];
  const int lane = threadIdx.x;
  const int nwx = p.Wp / WS;
  const int WY = blockIdx.x / nwx, WX = blockIdx.x - WY * nwx, hd = blockIdx.y;
  const int i = min(lane, NT - 1);
  const int wy = i / WS, wx = i - wy * WS;
  const int sy = WY * WS + wy, sx = WX * WS + wx;            // coordinates in the shifted map
  const int py = (sy + p.shift) % p.Hp, px = (sx + p.shift) % p.Wp;
  const bool valid = py < p.H && px < p.W;
  const long tok = (long)blockIdx.z * p.img_rows + (long)py * p.W + px;      // (blockIdx.z: image of the batch -- round 6: one launch for all)
  const float* row = p.qkv + tok * p.ld + hd * D;
  float q[D];
#pragma unroll
  for (int c4 = 0; c4 < D / 4; ++c4) {
    float4 tq, tk, tv;
    if (valid) {
      tq = *reinterpret_cast<const float4*>(row + c4 * 4);
      tk = *reinterpret_cast<const float4*>(row + p.C + c4 * 4);
      tv = *reinterpret_cast<const float4*>(row + 2 * p.C + c4 * 4);
    } else {
      tq = *reinterpret_cast<const float4*>(p.bias + hd * D + c4 * 4);
      tk = *reinterpret_cast<const float4*>(p.bias + p.C + hd * D + c4 * 4);
      tv = *reinterpret_cast<const float4*>(p.bias + 2 * p.C + hd * D + c4 * 4);
    }
    q[4 * c4] = tq.x * p.scale; q[4 * c4 + 1] = tq.y * p.scale; q[4 * c4 + 2] = tq.z * p.scale; q[4 * c4 + 3] = tq.w * p.scale;
    if (lane < NT) {
      *reinterpret_cast<float4*>(&Ks[lane][c4 * 4]) = tk;
      *reinterpret_cast<float4*>(&Vs[lane][c4 * 4]) = tv;
    }
  }
  for (int t = lane; t < 169; t += 64) tbl[t] = p.table[(long)t * p.nH + hd];
  // region label of the shifted-map position (BasicLayer.forward :395-407)
  const int ry = sy < p.Hp - WS ? 0 : (sy < p.Hp - WS / 2 ? 1 : 2);
  const int rx = sx < p.Wp - WS ? 0 : (sx < p.Wp - WS / 2 ? 1 : 2);
  const int mylab = ry * 3 + rx;
  if (lane < NT) lab[lane] = mylab;
  __syncthreads();

  float s[NT];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float dot = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < D / 4; ++c4) {
      const float4 kk = *reinterpret_cast<const float4*>(&Ks[j][c4 * 4]);
      dot = fmaf(q[4 * c4], kk.x, dot);
      dot = fmaf(q[4 * c4 + 1], kk.y, dot);
      dot = fmaf(q[4 * c4 + 2], kk.z, dot);
      dot = fmaf(q[4 * c4 + 3], kk.w, dot);
    }
    const int jy = j / WS, jx = j - jy * WS;
    dot += tbl[(wy - jy + WS - 1) * (2 * WS - 1) + (wx - jx + WS - 1)];
    if (p.shift > 0 && lab[j] != mylab) dot += -100.f;
    s[j] = dot;
    m = fmaxf(m, dot);
  }
  float l = 0.f;
  float o[D];
#pragma unroll
  for (int c = 0; c < D; ++c) o[c] = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float pw = expf(s[j] - m);
    l += pw;
#pragma unroll
    for (int c4 = 0; c4 < D / 4; ++c4) {
      const float4 vv = *reinterpret_cast<const float4*>(&Vs[j][c4 * 4]);
      o[4 * c4] = fmaf(pw, vv.x, o[4 * c4]);
      o[4 * c4 + 1] = fmaf(pw, vv.y, o[4 * c4 + 1]);
      o[4 * c4 + 2] = fmaf(pw, vv.z, o[4 * c4 + 2]);
      o[4 * c4 + 3] = fmaf(pw, vv.w, o[4 * c4 + 3]);
    }
  }
  if (lane < NT && valid) {
    const float inv = 1.f / l;
    float4* dst = reinterpret_cast<float4*>(p.out + tok * p.ldo + hd * D);
#pragma unroll
    for (int c4 = 0; c4 < D / 4; ++c4)
      dst[c4] = make_float4(o[4 * c4] * inv, o[4 * c4 + 1] * inv, o[4 * c4 + 2] * inv, o[4 * c4 + 3] * inv);
  }
}

extern "C" int aot_swin_window_attn_f32(const float* qkv, const float* qkv_bias, const float* rpb_table, float* out, int B, int H,
                                        int W, int C, int nH, int window, int shift, int ld, int ldo, float scale,
                                        void* stream) {
  if (!qkv || !qkv_bias || !rpb_table || !out || B <= 0 || B > 65535 || H <= 0 || W <= 0 || nH <= 0) return AOT_ERR_BADARG;
  if (window != 7 || C != nH * 32 || shift < 0 || shift >= window) return AOT_ERR_UNSUPPORTED;
  if ((ld & 3) || (ldo & 3) || ld < 3 * C || ldo < C) return AOT_ERR_BADARG;
  SwinAttnParams p;
  p.qkv = qkv; p.bias = qkv_bias; p.table = rpb_table; p.out = out;
  p.H = H; p.W = W; p.C = C; p.nH = nH; p.ld = ld; p.ldo = ldo; p.shift = shift;
  p.Hp = cdiv(H, 7) * 7; p.Wp = cdiv(W, 7) * 7; p.scale = scale;
  p.img_rows = (long)H * W;
  hipLaunchKernelGGL(swin_window_attn_kernel, dim3((p.Hp / 7) * (p.Wp / 7), nH, B), dim3(64), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}

// PatchMerging gather (:338-356): out[(Y,X)] = [x(2Y,2X) | x(2Y+1,2X) | x(2Y,2X+1) | x(2Y+1,2X+1)], zeros past the edge.
__global__ void __launch_bounds__(256) patch_merge_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W,
                                                          int C, int H2, int W2, int ldx) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nv = C >> 2;
  if (idx >= (long)H2 * W2 * 4 * nv) return;
  const int c4 = (int)(idx % nv);
  const int part = (int)((idx / nv) & 3);
  const int pix = (int)(idx / (4 * nv));
  const int Y = pix / W2, X = pix - Y * W2;
  const int y = 2 * Y + (part & 1), xx = 2 * X + (part >> 1);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (y < H && xx < W) v = *reinterpret_cast<const float4*>(x + ((long)y * W + xx) * ldx + c4 * 4);
  *reinterpret_cast<float4*>(out + (long)pix * 4 * C + part * C + c4 * 4) = v;
}

extern "C" int aot_patch_merge_f32(const float* x, float* out, int H, int W, int C, int ldx, void* stream) {
  if (!x || !out || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (ldx & 3)) return AOT_ERR_BADARG;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const long total = (long)H2 * W2 * C;
  hipLaunchKernelGGL(patch_merge_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, H, W, C, H2, W2, ldx);
  AOT_LAUNCH_CHECK();
}
