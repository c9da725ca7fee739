// Implicit-GEMM convolution / linear layer on the exact-fp32 matrix cores of gfx950: the register-staged LDS kernel, the
// wave-independent small-M kernels and the dispatcher (the LDS-direct tile kernel lives in gemm_lds.hip).
//
//   out[m, n] = act( sum_k A[m, k] * W[k, n] + bias[n] + res[m, n] )
//
// A is the im2col view of an NHWC activation map, generated on the fly by the loader (never
// materialised); W is the [KH*KW*Cin, Cout] weight with FrozenBN folded in by the host.
// MFMA: v_mfma_f32_32x32x2_f32 -- an exact fp32 fmaf chain at the fp32 vector rate (157 TF peak),
// so results are bit-comparable with an fp32 reference up to summation order.
//
// Tiling: a workgroup owns a BM x BN tile, each 64-lane wave a WM x WN sub-tile built from
// 32x32 MFMA fragments; K is walked in BK=16 slabs through a 2-stage LDS ring (global -> regs
// issued before the MFMA block, regs -> LDS after it, one barrier per slab).  LDS images:
//   As[k][m] (k-major, row stride BM+2: conflict-free ds_write_b32 of the transposed float4 and
//             conflict-free ds_read_b32 of the A fragment, lane i <-> row i)
//   Bs[k][n] (row stride BN+4, float4 stores, lane j <-> column j)
// fp32 MFMA issues once per 64 cycles per SIMD, so one ds_read_b32 per operand is far from the
// LDS limit (section 3 of the guide: 4 LDS cycles per 4 MFMAs = 256 cycles).
#include "conv_params.h"
#include <cstdlib>

// GNA (round 6, 1x1 only): the A operand is read THROUGH GroupNorm-apply + activation -- `in` is the un-normalised output of a ConvGN
// block and gna its finished statistics (per lane of gna.rows rows and group: mean, rstd in double) and affine parameters: each loaded
// float4 is normalised with gn_apply_kernel's own arithmetic, so the product equals gn_apply -> conv bit for bit while the normalised
// map is never written (the FPN head's conv_4x block feeds nothing but conv_out: fpn.py:56-58 in the reference).
struct GnApplyIn {
  const double* stats;   // [lanes][G][2]
  const float* gamma;
  const float* beta;
  int G, act, rows;      // rows per lane
};
__device__ __forceinline__ float gna_act(float v, int act) {      // (gn_act of norm_act.hip)
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 3) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  return v;
}
template <int BM, int BN, int WM, int WN, int BK, bool IS1X1, bool GNA = false>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64)
conv_gemm_kernel(const ConvParams p, const GnApplyIn gna) {
  static_assert(!GNA || IS1X1, "GroupNorm-apply on the A operand: a 1x1 layer");
  constexpr int NW_N = BN / WN;
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_F4 = BM * BK / 4;
  constexpr int B_F4 = BK * BN / 4;
  constexpr int A_PER = (A_F4 + NT - 1) / NT;
  constexpr int B_PER = (B_F4 + NT - 1) / NT;
  constexpr int KQ = BK / 4;                 // float4 per A row of a slab
  constexpr int LDA_S = BM + (BK == 16 ? 2 : 1);   // keeps the transposed ds_write_b32 conflict-free
  constexpr int LDB_S = BN + 4;

  __shared__ float As[2][BK][LDA_S];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB_S];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int nbn = (p.Cout + BN - 1) / BN;
  const int nbm = (p.M + BM - 1) / BM;
  // XCD-aware bijective remap: block b runs on XCD b%8; give each XCD a contiguous run of tiles
  // so tiles that share an A row-panel hit the same private L2.
  const int nwg = nbm * nbn, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int wm0 = (wave / NW_N) * WM, wn0 = (wave % NW_N) * WN;

  // ---- A loader state (one float4 = 4 consecutive k of one output pixel) ----
  bool a_ok[A_PER];
  long a_base[A_PER], a_img[A_PER];
  int a_iy0[A_PER], a_ix0[A_PER], a_c[A_PER], a_ky[A_PER], a_kx[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int f = tid + i * NT;
    const int r = f / KQ, kq = f % KQ;
    const int m = m0 + r;
    a_ok[i] = (f < A_F4) && (m < p.M);
    const int mm = a_ok[i] ? m : 0;
    const int hw_out = p.OH * p.OW;
    const int bi = mm / hw_out, pix = mm - bi * hw_out;     // image of the batch, output pixel
    const int oy = pix / p.OW, ox = pix - oy * p.OW;
    a_img[i] = (long)bi * p.H * p.W * p.lda;
    if (IS1X1) {
      a_base[i] = a_img[i] + ((long)(oy * p.stride) * p.W + ox * p.stride) * p.lda + kq * 4;
    } else {
      a_iy0[i] = oy * p.stride - p.pad;
      a_ix0[i] = ox * p.stride - p.pad;
      const int k = kq * 4;
      const int tap = k / p.Cin;
      a_c[i] = k - tap * p.Cin;
      a_ky[i] = tap / p.KW;
      a_kx[i] = tap - a_ky[i] * p.KW;
    }
  }

  float4 a_reg[A_PER], b_reg[B_PER];

  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int f = tid + i * NT;
      const int k = kt * BK + (f % KQ) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (IS1X1) {
        if (a_ok[i] && k < p.K) {
          v = *reinterpret_cast<const float4*>(p.in + a_base[i] + (long)kt * BK);
          if (GNA) {
            const int m = m0 + f / KQ;
            const double* st = gna.stats + ((long)(m / gna.rows) * gna.G + k / (p.K / gna.G)) * 2;
            const float mean = (float)st[0], rstd = (float)st[1];
            const float4 ga = *reinterpret_cast<const float4*>(gna.gamma + k), be = *reinterpret_cast<const float4*>(gna.beta + k);
            v.x = gna_act((v.x - mean) * rstd * ga.x + be.x, gna.act);
            v.y = gna_act((v.y - mean) * rstd * ga.y + be.y, gna.act);
            v.z = gna_act((v.z - mean) * rstd * ga.z + be.z, gna.act);
            v.w = gna_act((v.w - mean) * rstd * ga.w + be.w, gna.act);
          }
        }
      } else {
        const int iy = a_iy0[i] + a_ky[i] * p.dil, ix = a_ix0[i] + a_kx[i] * p.dil;
        if (a_ok[i] && k < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          v = *reinterpret_cast<const float4*>(p.in + a_img[i] + ((long)iy * p.W + ix) * p.lda + a_c[i]);
        // advance (tap, c) by BK channels-of-k
        a_c[i] += BK;
        while (a_c[i] >= p.Cin) {
          a_c[i] -= p.Cin;
          if (++a_kx[i] == p.KW) { a_kx[i] = 0; ++a_ky[i]; }
        }
      }
      a_reg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int f = tid + i * NT;
      const int kr = f / (BN / 4), n4 = f - kr * (BN / 4);
      const int k = kt * BK + kr, n = n0 + n4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < B_F4 && k < p.K && n < p.ldb) v = *reinterpret_cast<const float4*>(p.w + (long)k * p.ldb + n);
      b_reg[i] = v;
    }
  };

  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int f = tid + i * NT;
      if (f < A_F4) {
        const int r = f / KQ, kq = f % KQ;
        As[buf][kq * 4 + 0][r] = a_reg[i].x;
        As[buf][kq * 4 + 1][r] = a_reg[i].y;
        As[buf][kq * 4 + 2][r] = a_reg[i].z;
        As[buf][kq * 4 + 3][r] = a_reg[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int f = tid + i * NT;
      if (f < B_F4) {
        const int kr = f / (BN / 4), n4 = f - kr * (BN / 4);
        *reinterpret_cast<float4*>(&Bs[buf][kr][n4 * 4]) = b_reg[i];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);  // global loads fly under the MFMA block below
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][kk + kh][wm0 + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[buf][kk + kh][wn0 + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias (folded BN) + residual + activation; 32 lanes write 128 contiguous bytes ----
  with_act(p.act, [&](auto ACT) __attribute__((always_inline)) -> void {
    constexpr int act = decltype(ACT)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + l31;
        if (n >= p.Cout) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + i * 32 + mfma32_row(r, kh);
          if (m < p.M) {
            float v = acc[i][j][r] + bv;
            if (p.res) v += p.res[(long)(p.res_rows ? m % p.res_rows : m) * p.ldr + n];
            p.out[(long)m * p.ldc + n] = apply_act(v, act);
          }
        }
      }
  });
}


// ---------------------------------------------------------------------------------------------------------
// Small-M GEMM / conv (the stride-16 maps: M = 1674 at 480p).  A 64x64-tiled launch has only ~100 workgroups
// for 256 CUs and each walks K serially, so here every WAVE is independent: it owns a 32x32 output tile and
// a contiguous K range, loads its MFMA operands straight from global/L2 into registers (A: 16 consecutive k of
// its row as 4 x float4, the contraction index being enumerated as k = slab + 16*half + s for both operands;
// B: one coalesced 128-byte row segment per k), and never touches a barrier in the main loop.  KS waves of a
// workgroup split K; their accumulators meet in LDS once, are summed in fixed wave order (deterministic) and
// each wave finishes 16/KS of the fragment rows with the bias/residual/activation epilogue.
// Requires Cin % 32 == 0 (a 32-wide k slab never straddles a filter tap).
// ---------------------------------------------------------------------------------------------------------
template <int KS, bool IS1X1>
__global__ void __launch_bounds__(KS * 64) gemm_direct_kernel(const ConvParams p) {
  __shared__ float red[KS > 1 ? KS : 1][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, kh = lane >> 5;
  const int nbn = (p.Cout + 31) >> 5, nbm = (p.M + 31) >> 5;
  const int nwg = nbm * nbn, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int m0 = bm * 32, n0 = bn * 32;

  const int nslab = p.K >> 5;
  const int per = (nslab + KS - 1) / KS;
  const int s0 = wave * per, s1 = min(nslab, s0 + per);

  const int m = m0 + j;
  const bool row_ok = m < p.M;
  const int mm = row_ok ? m : 0;
  const int hw_out = p.OH * p.OW;
  const int bi = mm / hw_out, pix = mm - bi * hw_out;       // image of the batch, output pixel
  const int oy = pix / p.OW, ox = pix - oy * p.OW;
  const float* img = p.in + (long)bi * p.H * p.W * p.lda;
  long a_base = 0;
  int iy0 = 0, ix0 = 0, c = 0, ky = 0, kx = 0;
  if (IS1X1) {
    a_base = ((long)(oy * p.stride) * p.W + ox * p.stride) * p.lda + kh * 16;
  } else {
    iy0 = oy * p.stride - p.pad;
    ix0 = ox * p.stride - p.pad;
    const int k = s0 * 32;
    const int tap = k / p.Cin;
    c = k - tap * p.Cin;
    ky = tap / p.KW;
    kx = tap - ky * p.KW;
  }
  const int n = n0 + j;
  const bool col_ok = n < p.ldb;
  const float* bcol = p.w + (col_ok ? n : 0);

  float4 an[4];
  float bn_[16];
  auto load = [&](int s) {
    const float* src = nullptr;
    if (IS1X1) {
      if (row_ok) src = img + a_base + (long)s * 32;
    } else {
      const int iy = iy0 + ky * p.dil, ix = ix0 + kx * p.dil;
      if (row_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        src = img + ((long)iy * p.W + ix) * p.lda + c + kh * 16;
      c += 32;
      if (c >= p.Cin) { c = 0; if (++kx == p.KW) { kx = 0; ++ky; } }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) an[i] = src ? reinterpret_cast<const float4*>(src)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* bp = bcol + ((long)s * 32 + kh * 16) * p.ldb;
#pragma unroll
    for (int t = 0; t < 16; ++t) bn_[t] = col_ok ? bp[(long)t * p.ldb] : 0.f;
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (s0 < s1) load(s0);
  for (int s = s0; s < s1; ++s) {
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[4 * i] = an[i].x; a[4 * i + 1] = an[i].y; a[4 * i + 2] = an[i].z; a[4 * i + 3] = an[i].w; }
#pragma unroll
    for (int t = 0; t < 16; ++t) b[t] = bn_[t];
    if (s + 1 < s1) load(s + 1);
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
  }

  constexpr int RPW = 16 / KS;   // fragment registers finished by each wave
  float fin[RPW];
  if (KS > 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave * RPW + i;
      float v = red[0][r][lane];
#pragma unroll
      for (int w = 1; w < KS; ++w) v += red[w][r][lane];
      fin[i] = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < RPW; ++i) fin[i] = acc[i];
  }
  if (n < p.Cout) {
    const float bv = p.bias ? p.bias[n] : 0.f;
    with_act(p.act, [&](auto ACT) __attribute__((always_inline)) -> void {
      constexpr int act = decltype(ACT)::value;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = (KS > 1 ? wave * RPW : 0) + i;
      const int mo = m0 + mfma32_row(r, kh);
      if (mo < p.M) {
        float v = fin[i] + bv;
        if (p.res) v += p.res[(long)(p.res_rows ? mo % p.res_rows : mo) * p.ldr + n];
        p.out[(long)mo * p.ldc + n] = apply_act(v, act);
      }
    }
    });
  }
}

template <int KS>
static int launch_direct(const ConvParams& p, bool is1x1, hipStream_t s) {
  const int nb = cdiv(p.M, 32) * cdiv(p.Cout, 32);
  if (is1x1)
    hipLaunchKernelGGL((gemm_direct_kernel<KS, true>), dim3(nb), dim3(KS * 64), 0, s, p);
  else
    hipLaunchKernelGGL((gemm_direct_kernel<KS, false>), dim3(nb), dim3(KS * 64), 0, s, p);
  AOT_LAUNCH_CHECK();
}


// Second generation of the wave-independent kernel: a wave owns a 64x32 output tile (two 32x32 fragments that
// share the B operand -> two independent MFMA chains and half the B loads per MFMA), the register sets of
// consecutive k slabs ping-pong (no copies), and B rows come through a buffer descriptor whose row offset is a
// wave-uniform scalar (no address VALU; columns past ldb read 0 by the hardware bounds check).
template <int KS, bool IS1X1>
__global__ void __launch_bounds__(KS * 64) gemm_direct2_kernel(const ConvParams p) {
  __shared__ float red[KS > 1 ? KS : 1][32][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, kh = lane >> 5;
  const int nbn = (p.Cout + 31) >> 5, nbm = (p.M + 63) >> 6;
  const int nwg = nbm * nbn, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int m0 = bm * 64, n0 = bn * 32;
  const int nslab = p.K >> 5;
  const int per = (nslab + KS - 1) / KS;
  const int s0 = wave * per, s1 = min(nslab, s0 + per);

  bool row_ok[2];
  const float* a1x1[2];
  const float* img2[2];
  int iy0[2], ix0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + i * 32 + j;
    row_ok[i] = m < p.M;
    const int mm = row_ok[i] ? m : 0;
    const int hw_out = p.OH * p.OW;
    const int bi = mm / hw_out, pix = mm - bi * hw_out;
    const int oy = pix / p.OW, ox = pix - oy * p.OW;
    img2[i] = p.in + (long)bi * p.H * p.W * p.lda;
    a1x1[i] = img2[i] + ((long)(oy * p.stride) * p.W + ox * p.stride) * p.lda + kh * 16 + (long)s0 * 32;
    iy0[i] = oy * p.stride - p.pad;
    ix0[i] = ox * p.stride - p.pad;
  }
  int c = 0, ky = 0, kx = 0;   // filter tap / channel of the NEXT slab to load (wave-uniform)
  if (!IS1X1) {
    const int k = s0 * 32;
    const int tap = k / p.Cin;
    c = k - tap * p.Cin;
    ky = tap / p.KW;
    kx = tap - ky * p.KW;
  }
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.K * p.ldb * 4, 0x00020000);
  const int n = n0 + j;
  const int bvoff = (n < p.ldb) ? (kh * 16 * p.ldb + n) * 4 : 0x7ffffff0;   // out-of-range columns read 0
  const int ldb4 = p.ldb * 4;

  auto load = [&](float4 (&a)[2][4], float (&b)[16], int s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* src = nullptr;
      if (IS1X1) {
        if (row_ok[i]) src = a1x1[i];
        a1x1[i] += 32;
      } else {
        const int iy = iy0[i] + ky * p.dil, ix = ix0[i] + kx * p.dil;
        if (row_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          src = img2[i] + ((long)iy * p.W + ix) * p.lda + c + kh * 16;
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) a[i][v] = src ? reinterpret_cast<const float4*>(src)[v] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!IS1X1) {
      c += 32;
      if (c >= p.Cin) { c = 0; if (++kx == p.KW) { kx = 0; ++ky; } }
    }
    const int so = s * 32 * ldb4;
#pragma unroll
    for (int t = 0; t < 16; ++t)
      b[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, bvoff, so + t * ldb4, 0));
  };

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  auto mma = [&](const float4 (&a)[2][4], const float (&b)[16]) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][v].x, b[4 * v], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][v].x, b[4 * v], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][v].y, b[4 * v + 1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][v].y, b[4 * v + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][v].z, b[4 * v + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][v].z, b[4 * v + 2], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][v].w, b[4 * v + 3], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][v].w, b[4 * v + 3], acc1, 0, 0, 0);
    }
  };
  float4 aA[2][4], aB[2][4];
  float bA[16], bB[16];
  if (s0 < s1) load(aA, bA, s0);
  for (int s = s0; s < s1; s += 2) {
    if (s + 1 < s1) load(aB, bB, s + 1);
    mma(aA, bA);
    if (s + 1 < s1) {
      if (s + 2 < s1) load(aA, bA, s + 2);
      mma(aB, bB);
    }
  }

  // ---- in-block split-K reduction (fixed wave order) + epilogue: wave w finishes 32/KS of the 32 fragment rows ----
  constexpr int RPW = 32 / KS;
  float fin[RPW];
  if (KS > 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { red[wave][r][lane] = acc0[r]; red[wave][16 + r][lane] = acc1[r]; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave * RPW + i;
      float v = red[0][r][lane];
#pragma unroll
      for (int w = 1; w < KS; ++w) v += red[w][r][lane];
      fin[i] = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) { fin[i] = acc0[i]; fin[16 + i] = acc1[i]; }
  }
  if (n < p.Cout) {
    const float bv = p.bias ? p.bias[n] : 0.f;
    with_act(p.act, [&](auto ACT) __attribute__((always_inline)) -> void {
      constexpr int act = decltype(ACT)::value;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = (KS > 1 ? wave * RPW : 0) + i;      // 0..15 -> fragment 0, 16..31 -> fragment 1
      const int mo = m0 + (r >> 4) * 32 + mfma32_row(r & 15, kh);
      if (mo < p.M) {
        float v = fin[i] + bv;
        if (p.res) v += p.res[(long)(p.res_rows ? mo % p.res_rows : mo) * p.ldr + n];
        p.out[(long)mo * p.ldc + n] = apply_act(v, act);
      }
    }
    });
  }
}

template <int KS>
static int launch_direct2(const ConvParams& p, bool is1x1, hipStream_t s) {
  const int nb = cdiv(p.M, 64) * cdiv(p.Cout, 32);
  if (is1x1)
    hipLaunchKernelGGL((gemm_direct2_kernel<KS, true>), dim3(nb), dim3(KS * 64), 0, s, p);
  else
    hipLaunchKernelGGL((gemm_direct2_kernel<KS, false>), dim3(nb), dim3(KS * 64), 0, s, p);
  AOT_LAUNCH_CHECK();
}

template <int BM, int BN, int WM, int WN, int BK>
static int launch_cfg(const ConvParams& p, bool is1x1, hipStream_t s) {
  const int nb = cdiv(p.M, BM) * cdiv(p.Cout, BN);
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  if (is1x1)
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, BK, true>), dim3(nb), dim3(NT), 0, s, p, GnApplyIn{});
  else
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, BK, false>), dim3(nb), dim3(NT), 0, s, p, GnApplyIn{});
  AOT_LAUNCH_CHECK();
}

// Kernel choice (cfg < 0): a table read off the sweeps of tools/dev/gemm_check.hip on MI355X (profiles/r02_gemm_sweep.txt,
// profiles/r02f_gemm_sweep_lean.txt; every kernel family on every conv / linear shape of the R50-AOTL frame, one clip and
// three lanes stacked):
//   * lean LDS-direct 64x64 tile kernel (gemm_lds.hip, cfg 197): everything with Cin % 32 == 0 and >= 192 tiles of 64x64 --
//     15-25 % faster than both older tile kernels on every such shape (buffer-DMA addressing, fragment reads and DMA
//     pieces placed between the MFMAs);
//   * register-staged 64x64x32 LDS kernel (cfg 4): what the lean kernel cannot take (stem, MobileNet channel counts);
//     128x32 tiles for Cout <= 32;
//   * wave-independent kernels with in-block split-K (cfg 1x: 32x32 waves, cfg 2x: 64x32 waves): the stride-16 maps
//     (M = 1674 per lane) with fewer than 192 tiles, where no LDS-tiled kernel -- with or without split-K slabs --
//     beats them.
// Two tables.  LATENCY (cfg -1): the fastest kernel for each shape run on its own -- one clip at a time.  THROUGHPUT (cfg -2):
// the kernel that costs the least SIMD time when several clips keep the chip saturated -- the wave-independent kernels win
// the stride-16 shapes in isolation (more workgroups), but issue twice the VALU instructions per MFMA, and VALU work does not
// hide under fp32 MFMAs: with three clips per GPU the lean tile kernel everywhere is +4..14 % frames/s.
static int auto_cfg(const ConvParams& p, long scratch_floats, bool throughput) {
  (void)scratch_floats;
  if (!p.w) return gemm_lean_eligible(p) ? 196 + 1 : 4;      // only the k-contiguous twin given: the LDS-direct kernel or nothing
  const bool direct_ok = (p.Cin % 32 == 0) && (p.K % 32 == 0);
  const long tiles64 = (long)cdiv(p.M, 64) * cdiv(p.Cout, 64);
  const bool kxk = p.KH * p.KW > 1;
  if (p.Cout <= 32) return 3;
  if (tiles64 >= (throughput ? 16 : 192) && gemm_lean_eligible(p)) return 196 + 1;
  if (tiles64 >= 512 || !direct_ok) return 4;
  if (kxk && tiles64 < 192 && p.K >= 1024) return 24;
  const long tiles32 = (long)cdiv(p.M, 32) * cdiv(p.Cout, 32);
  const int nslab = p.K / 32;
  int ks = 1;
  while (ks < 8 && tiles32 * ks < 2048 && nslab / (ks * 2) >= 2) ks *= 2;
  return 10 + ks;
}

extern "C" int aot_conv2d_nhwc_f32(const float* in, const float* w, const float* wt, const float* bias, const float* res,
                                   float* out, float* scratch, long scratch_floats, int B, int H, int W, int Cin, int OH,
                                   int OW, int Cout, int KH, int KW, int stride, int pad, int dil, int lda, int ldb,
                                   int ldwt, int ldc, int ldr, int res_rows, int act, int cfg, void* stream) {
  if (!in || (!w && !wt) || !out) return AOT_ERR_BADARG;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || OH <= 0 || OW <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return AOT_ERR_BADARG;
  if ((Cin & 3) || (lda & 3) || lda < Cin || ldc < Cout) return AOT_ERR_BADARG;
  if (w && ((ldb & 3) || ldb < Cout || ((uintptr_t)w & 15))) return AOT_ERR_BADARG;
  if ((uintptr_t)in & 15) return AOT_ERR_BADARG;
  if (res && (ldr < Cout || res_rows < 0)) return AOT_ERR_BADARG;
  if ((long)B * OH * OW > 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  ConvParams p;
  p.in = in; p.w = w; p.wt = wt; p.bias = bias; p.res = res; p.out = out;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.lda = lda; p.ldb = ldb; p.ldwt = ldwt; p.ldc = ldc; p.ldr = ldr; p.res_rows = res_rows;
  p.M = B * OH * OW; p.K = KH * KW * Cin; p.act = act;
  if (wt && ldwt < p.K) return AOT_ERR_BADARG;
  const bool is1x1 = (KH == 1 && KW == 1 && pad == 0);
  hipStream_t s = (hipStream_t)stream;
  if (!scratch) scratch_floats = 0;
  if (cfg < 0) cfg = auto_cfg(p, scratch_floats, cfg == -2);
  if (cfg < 100 && !w) return AOT_ERR_UNSUPPORTED;      // every kernel but the LDS-direct ones reads w [K, ldb]
  if (cfg >= 100) {      // LDS-direct kernel: variant (cfg - 100) / 16, split-K factor (cfg - 100) % 16
    const int variant = (cfg - 100) / 16, ks = (cfg - 100) % 16;
    if (ks > 1 && (long)ks * p.M * p.Cout > scratch_floats) return AOT_ERR_BADARG;
    return launch_gemm_lds(p, variant, ks, scratch, s);
  }
  switch (cfg) {
    case 0: return launch_cfg<128, 128, 64, 64, 16>(p, is1x1, s);
    case 1: return launch_cfg<128, 64, 64, 32, 16>(p, is1x1, s);
    case 2: return launch_cfg<64, 64, 32, 32, 16>(p, is1x1, s);
    case 3: return launch_cfg<128, 32, 32, 32, 16>(p, is1x1, s);
    case 4: return launch_cfg<64, 64, 32, 32, 32>(p, is1x1, s);
    case 5: return launch_cfg<128, 64, 64, 32, 32>(p, is1x1, s);
    case 11: case 12: case 14: case 18: case 21: case 22: case 24: case 28:
      if ((Cin % 32) || (p.K % 32)) return AOT_ERR_UNSUPPORTED;
      switch (cfg) {
        case 11: return launch_direct<1>(p, is1x1, s);
        case 12: return launch_direct<2>(p, is1x1, s);
        case 14: return launch_direct<4>(p, is1x1, s);
        case 18: return launch_direct<8>(p, is1x1, s);
        case 21: return launch_direct2<1>(p, is1x1, s);
        case 22: return launch_direct2<2>(p, is1x1, s);
        case 24: return launch_direct2<4>(p, is1x1, s);
        default: return launch_direct2<8>(p, is1x1, s);
      }
    default: return AOT_ERR_BADARG;
  }
}

// out = act(conv1x1(gn_act(GroupNorm(in))) + bias) for Cout <= 32 (the FPN head's conv_out behind its conv_4x block): GroupNorm-apply
// + activation folded into the A loads of the 128x32 register-staged kernel -- bit-identical to aot_groupnorm_apply_f32 followed by
// aot_conv2d_nhwc_f32 (cfg 3).  in [B*M, lda] un-normalised, stats [B][G][2] doubles, w [K, ldb] k-major.
extern "C" int aot_gn_conv1x1_f32(const float* in, const double* stats, const float* gamma, const float* beta, const float* w,
                                  const float* bias, float* out, int B, int M, int K, int Cout, int G, int lda, int ldb, int ldc,
                                  int gn_act, int act, void* stream) {
  if (!in || !stats || !gamma || !beta || !w || !out || B <= 0 || M <= 0 || K <= 0 || Cout <= 0 || G <= 0) return AOT_ERR_BADARG;
  if ((K & 3) || (lda & 3) || lda < K || ldc < Cout || (ldb & 3) || ldb < Cout || ((uintptr_t)w & 15) || ((uintptr_t)in & 15) ||
      ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15))
    return AOT_ERR_BADARG;
  if (K % G || ((K / G) & 3)) return AOT_ERR_BADARG;
  if (Cout > 32 || (long)B * M > 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  ConvParams p;
  p.in = in; p.w = w; p.wt = nullptr; p.bias = bias; p.res = nullptr; p.out = out;
  p.B = 1; p.H = 1; p.W = B * M; p.Cin = K; p.OH = 1; p.OW = B * M; p.Cout = Cout;
  p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.lda = lda; p.ldb = ldb; p.ldwt = 0; p.ldc = ldc; p.ldr = 0; p.res_rows = 0;
  p.M = B * M; p.K = K; p.act = act;
  GnApplyIn gna;
  gna.stats = stats; gna.gamma = gamma; gna.beta = beta; gna.G = G; gna.act = gn_act; gna.rows = M;
  const int nb = cdiv(p.M, 128) * cdiv(p.Cout, 32);
  hipLaunchKernelGGL((conv_gemm_kernel<128, 32, 32, 32, 16, true, true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, p, gna);
  AOT_LAUNCH_CHECK();
}

// ---- bf16 x 6 family (gemm_lds.hip: gemm_x6rd_kernel and friends) --------------------------------------------------------------
// weight [K, ldb] fp32 -> three truncated-bf16 planes in the kernel's tile order, w6[plane][K/32][cc][cout_pad][8]:
// element e of chunk cc = 2 s + h of k-block kb is k = 32 kb + 16 s + 4 h + (e & 3) + 8 (e >> 2); columns >= Cout are zero.
// One thread per (kb, cc, n): reads 8 weights, writes three 16-byte chunks.
__global__ void __launch_bounds__(256) pack_x6_kernel(const float* __restrict__ w, unsigned* __restrict__ w6, int K, int Cout,
                                                       int ldb, int cout_pad) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)(K / 32) * 4 * cout_pad;
  if (idx >= total) return;
  const int n = (int)(idx % cout_pad);
  const int cc = (int)((idx / cout_pad) & 3);
  const int kb = (int)(idx / ((long)cout_pad * 4));
  const int k0 = 32 * kb + 16 * (cc >> 1) + 4 * (cc & 1);
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = n < Cout ? w[(long)(k0 + (e & 3) + 8 * (e >> 2)) * ldb + n] : 0.f;
  const long plane = (long)(K / 8) * cout_pad * 4;      // dwords per plane
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned a = __float_as_uint(x[2 * e]) & 0xffff0000u, b = __float_as_uint(x[2 * e + 1]) & 0xffff0000u;
      o[e] = (a >> 16) | b;
      x[2 * e] -= __uint_as_float(a);       // exact: the remainder of a truncation
      x[2 * e + 1] -= __uint_as_float(b);
    }
    *reinterpret_cast<uint4*>(w6 + pl * plane + idx * 4) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// one plane, rounded to nearest even: the weight operand of the plain-bf16 (training) form, same tile order as plane 0 above
__global__ void __launch_bounds__(256) pack_bf16_kernel(const float* __restrict__ w, unsigned* __restrict__ wq, int K, int Cout, int ldb,
                                                        int cout_pad) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)(K / 32) * 4 * cout_pad;
  if (idx >= total) return;
  const int n = (int)(idx % cout_pad);
  const int cc = (int)((idx / cout_pad) & 3);
  const int kb = (int)(idx / ((long)cout_pad * 4));
  const int k0 = 32 * kb + 16 * (cc >> 1) + 4 * (cc & 1);
  unsigned o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned h[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ee = 2 * e + t;
      const unsigned u = n < Cout ? __float_as_uint(w[(long)(k0 + (ee & 3) + 8 * (ee >> 2)) * ldb + n]) : 0u;
      // round to nearest even on the dropped 16 bits; NaN / Inf keep their exponent (a quiet NaN stays a NaN)
      h[t] = ((u & 0x7f800000u) == 0x7f800000u) ? (u >> 16) | ((u & 0xffffu) ? 0x40u : 0u) : (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    }
    o[e] = h[0] | (h[1] << 16);
  }
  *reinterpret_cast<uint4*>(wq + idx * 4) = make_uint4(o[0], o[1], o[2], o[3]);
}

extern "C" int aot_pack_bf16_f32(const float* w, void* wq, int K, int Cout, int ldb, int cout_pad, void* stream) {
  if (!w || !wq || K <= 0 || (K % 32) || Cout <= 0 || ldb < Cout || cout_pad < Cout || (cout_pad % 64) || ((uintptr_t)wq & 15))
    return AOT_ERR_BADARG;
  const long total = (long)(K / 32) * 4 * cout_pad;
  hipLaunchKernelGGL(pack_bf16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (unsigned*)wq, K, Cout, ldb, cout_pad);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_pack_bf16x6_f32(const float* w, void* w6, int K, int Cout, int ldb, int cout_pad, void* stream) {
  if (!w || !w6 || K <= 0 || (K % 32) || Cout <= 0 || ldb < Cout || cout_pad < Cout || (cout_pad % 64) || ((uintptr_t)w6 & 15))
    return AOT_ERR_BADARG;
  const long total = (long)(K / 32) * 4 * cout_pad;
  hipLaunchKernelGGL(pack_x6_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (unsigned*)w6, K, Cout, ldb,
                     cout_pad);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_conv2d_bf16x6_f32(const float* in, const void* w6, int cout_pad, const float* bias, const float* res,
                                     float* out, int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
                                     int stride, int pad, int dil, int lda, int ldc, int ldr, int res_rows, int act,
                                     int tile, void* stream) {
  if (!in || !w6 || !out || (tile != 0 && tile != 66 && tile != 129)) return AOT_ERR_BADARG;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || OH <= 0 || OW <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return AOT_ERR_BADARG;
  if ((lda & 3) || lda < Cin || ldc < Cout) return AOT_ERR_BADARG;
  if (res && (ldr < Cout || res_rows < 0)) return AOT_ERR_BADARG;
  if ((long)B * OH * OW > 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  ConvParams p;
  p.in = in; p.w = nullptr; p.wt = nullptr; p.bias = bias; p.res = res; p.out = out;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.lda = lda; p.ldb = 0; p.ldwt = 0; p.ldc = ldc; p.ldr = ldr; p.res_rows = res_rows;
  p.M = B * OH * OW; p.K = KH * KW * Cin; p.act = act;
  return launch_gemm_x6(p, w6, cout_pad, tile, (hipStream_t)stream);
}

// the phase-shifted 128x128 form with split-K over the grid (gemm_x6pp_kernel<., true>): partial tiles to `scratch`
// ([ksplit][M][Cout] floats), summed in slice order (+ bias / residual / act) by splitk_reduce_kernel
extern "C" int aot_conv2d_bf16x6k_f32(const float* in, const void* w6, int cout_pad, const float* bias, const float* res,
                                      float* out, int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
                                      int stride, int pad, int dil, int lda, int ldc, int ldr, int res_rows, int act,
                                      int ksplit, float* scratch, long scratch_floats, void* stream) {
  // ksplit < 0: |ksplit| slices on the 64x64 register-staged kernel with direct weight fragments (gemm_x6rd_kernel<., true>)
  const int ks = ksplit < 0 ? -ksplit : ksplit;
  if (!in || !w6 || !out || ks < 2 || ks > 64) return AOT_ERR_BADARG;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || OH <= 0 || OW <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return AOT_ERR_BADARG;
  if ((lda & 3) || lda < Cin || ldc < Cout) return AOT_ERR_BADARG;
  if (res && (ldr < Cout || res_rows < 0)) return AOT_ERR_BADARG;
  if ((long)B * OH * OW > 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  if (ks > 1 && (!scratch || (long)ks * B * OH * OW * Cout > scratch_floats)) return AOT_ERR_BADARG;
  ConvParams p;
  p.in = in; p.w = nullptr; p.wt = nullptr; p.bias = bias; p.res = res; p.out = out;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.lda = lda; p.ldb = 0; p.ldwt = 0; p.ldc = ldc; p.ldr = ldr; p.res_rows = res_rows;
  p.M = B * OH * OW; p.K = KH * KW * Cin; p.act = act;
  if (ksplit < 0) return launch_gemm_x6rd_splitk(p, w6, cout_pad, (hipStream_t)stream, -ksplit, scratch);     // 64x64 form
  return launch_gemm_x6pp(p, w6, cout_pad, (hipStream_t)stream, ksplit, scratch);
}

// aot_conv2d_bf16x6k_f32 as a linear layer with Cout == 256 whose reduce launch also writes LayerNorm(out) to ln_out [M, ld_ln] (round 6:
// linear2 + residual of an LSTT block followed by the stack's output norm) -- bit-identical to aot_layernorm_f32 on the stored result
extern "C" int aot_linear_bf16x6k_ln_f32(const float* in, const void* w6, int cout_pad, const float* bias, const float* res, float* out,
                                         int M, int K, int Cout, int lda, int ldc, int ldr, int res_rows, int act, int ksplit,
                                         float* scratch, long scratch_floats, const float* ln_gamma, const float* ln_beta, float* ln_out,
                                         int ld_ln, float eps, void* stream) {
  const int ks = ksplit < 0 ? -ksplit : ksplit;
  if (!in || !w6 || !out || ks < 2 || ks > 64 || !ln_gamma || !ln_beta || !ln_out || !(eps > 0.f)) return AOT_ERR_BADARG;
  if (M <= 0 || K <= 0 || (lda & 3) || lda < K || ldc < Cout || (ld_ln & 3) || ld_ln < Cout) return AOT_ERR_BADARG;
  if (res && (ldr < Cout || res_rows < 0)) return AOT_ERR_BADARG;
  if (!scratch || (long)ks * M * Cout > scratch_floats) return AOT_ERR_BADARG;
  if (Cout != 256 || ((uintptr_t)ln_out & 15) || ((uintptr_t)ln_gamma & 15) || ((uintptr_t)ln_beta & 15)) return AOT_ERR_UNSUPPORTED;
  ConvParams p;
  p.in = in; p.w = nullptr; p.wt = nullptr; p.bias = bias; p.res = res; p.out = out;
  p.B = 1; p.H = 1; p.W = M; p.Cin = K; p.OH = 1; p.OW = M; p.Cout = Cout;
  p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.lda = lda; p.ldb = 0; p.ldwt = 0; p.ldc = ldc; p.ldr = ldr; p.res_rows = res_rows;
  p.M = M; p.K = K; p.act = act;
  LnOutArgs ln;
  ln.gamma = ln_gamma; ln.beta = ln_beta; ln.out = ln_out; ln.ld = ld_ln; ln.eps = eps;
  if (ksplit < 0) return launch_gemm_x6rd_splitk(p, w6, cout_pad, (hipStream_t)stream, -ksplit, scratch, &ln);
  return launch_gemm_x6pp(p, w6, cout_pad, (hipStream_t)stream, ksplit, scratch, &ln);
}

// out = act(x W + bias (+ res)) on the bf16x6 family AND the GroupNorm partial sums of `out` (32-channel groups) from the same tile
// end: gn_part [2 * ceil(M / 64)][Cout / 32][2] floats (sum, sum of squares per 32-row block and group; gn_part_floats = its size)
extern "C" int aot_linear_gn_bf16x6_f32(const float* in, const void* w6, int cout_pad, const float* bias, const float* res, float* out,
                                        int M, int K, int Cout, int lda, int ldc, int ldr, int res_rows, int act, float* gn_part,
                                        long gn_part_floats, void* stream) {
  if (!in || !w6 || !out || !gn_part || M <= 0 || K <= 0 || Cout <= 0) return AOT_ERR_BADARG;
  if ((lda & 3) || lda < K || ldc < Cout || (K % 32) || (Cout % 32)) return AOT_ERR_BADARG;
  if (res && (ldr < Cout || res_rows < 0)) return AOT_ERR_BADARG;
  if (gn_part_floats < 2L * ((M + 63) / 64) * (Cout / 32) * 2) return AOT_ERR_BADARG;
  ConvParams p;
  p.in = in; p.w = nullptr; p.wt = nullptr; p.bias = bias; p.res = res; p.out = out;
  p.B = 1; p.H = 1; p.W = M; p.Cin = K; p.OH = 1; p.OW = M; p.Cout = Cout;
  p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.lda = lda; p.ldb = 0; p.ldwt = 0; p.ldc = ldc; p.ldr = ldr; p.res_rows = res_rows;
  p.M = M; p.K = K; p.act = act;
  return launch_gemm_x6rd_gn(p, w6, cout_pad, (hipStream_t)stream, gn_part);
}

// n <= 4 independent linear layers of one shape in ONE launch of the bf16x6 family (round 6): out[g] = act(in[g] W[g] + bias[g] (+ res[g])).
// The pointer arrays are HOST arrays read at launch time (bias / res: NULL, or an array whose entries are all set or all NULL).
extern "C" int aot_linear_group_bf16x6_f32(int n, const float* const* in, const void* const* w6, int cout_pad, const float* const* bias,
                                           const float* const* res, float* const* out, int M, int K, int Cout, int lda, int ldc, int ldr,
                                           int res_rows, int act, void* stream) {
  if (n < 1 || n > 4 || !in || !w6 || !out || M <= 0 || K <= 0 || Cout <= 0) return AOT_ERR_BADARG;
  if ((lda & 3) || lda < K || ldc < Cout || (K % 32)) return AOT_ERR_BADARG;
  if (res && res[0] && (ldr < Cout || res_rows < 0)) return AOT_ERR_BADARG;
  ConvParams p;
  p.in = in[0]; p.w = nullptr; p.wt = nullptr; p.bias = bias ? bias[0] : nullptr; p.res = res ? res[0] : nullptr; p.out = out[0];
  p.B = 1; p.H = 1; p.W = M; p.Cin = K; p.OH = 1; p.OW = M; p.Cout = Cout;
  p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.lda = lda; p.ldb = 0; p.ldwt = 0; p.ldc = ldc; p.ldr = ldr; p.res_rows = res_rows;
  p.M = M; p.K = K; p.act = act;
  return launch_gemm_x6rd_group(p, n, in, w6, bias, res, out, cout_pad, (hipStream_t)stream);
}

// out = act(LayerNorm(x) W + bias (+ res)) in ONE launch on the bf16x6 family (round 6; SURVEY 8b's aot_layernorm_linear, reference
// transformer.py:321-323 norm1 -> linear_Q|K|V and :355-359 norm3 -> linear1): x [M, lda] un-normalised, w6 = aot_pack_bf16x6_f32 of
// W' = diag(gamma) W, bias = beta W + b, colsum [Cout] = the column sums of W' -- all folded by the caller -- so the kernel normalises rows
// only (statistics along the k-loop, correction at the tile end: gemm_x6.hip); eps as nn.LayerNorm.  gn_part (optional): the GroupNorm
// partials of `out` as aot_linear_gn_bf16x6_f32 writes them.
extern "C" int aot_layernorm_linear_bf16x6_f32(const float* in, const void* w6, int cout_pad, const float* bias, const float* colsum,
                                               const float* res, float* out, int M, int K, int Cout, int lda, int ldc, int ldr,
                                               int res_rows, int act, float eps, float* gn_part, long gn_part_floats, void* stream) {
  if (!in || !w6 || !colsum || !out || M <= 0 || K <= 0 || Cout <= 0 || !(eps > 0.f)) return AOT_ERR_BADARG;
  if ((lda & 3) || lda < K || ldc < Cout || (K % 32)) return AOT_ERR_BADARG;
  if (res && (ldr < Cout || res_rows < 0)) return AOT_ERR_BADARG;
  if (gn_part && ((Cout % 32) || gn_part_floats < 2L * ((M + 63) / 64) * (Cout / 32) * 2)) return AOT_ERR_BADARG;
  ConvParams p;
  p.in = in; p.w = nullptr; p.wt = nullptr; p.bias = bias; p.res = res; p.out = out;
  p.B = 1; p.H = 1; p.W = M; p.Cin = K; p.OH = 1; p.OW = M; p.Cout = Cout;
  p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.lda = lda; p.ldb = 0; p.ldwt = 0; p.ldc = ldc; p.ldr = ldr; p.res_rows = res_rows;
  p.M = M; p.K = K; p.act = act;
  return launch_gemm_x6rd_ln(p, w6, cout_pad, (hipStream_t)stream, eps, colsum, gn_part);
}

// KxK convolution of B four-channel NHWC images (the ResNet stem: the image padded to r, g, b, 0) in the bf16x6 family: one 16-byte chunk
// of an im2col row = one filter tap, eight taps per k-step; w6 = aot_pack_bf16x6_f32 of the weight [Kp, ldb] with rows k = 4 * tap +
// channel and Kp = ceil(KH * KW / 8) * 32 (zero rows past KH * KW * 4)
extern "C" int aot_conv2d_c4_bf16x6_f32(const float* in, const void* w6, int cout_pad, const float* bias, float* out, int B, int H, int W,
                                        int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dil, int ldc, int act,
                                        void* stream) {
  if (!in || !w6 || !out || B <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || ldc < Cout)
    return AOT_ERR_BADARG;
  if ((long)B * OH * OW > 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  ConvParams p;
  p.in = in; p.w = nullptr; p.wt = nullptr; p.bias = bias; p.res = nullptr; p.out = out;
  p.B = B; p.H = H; p.W = W; p.Cin = 4; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.lda = 4; p.ldb = 0; p.ldwt = 0; p.ldc = ldc; p.ldr = 0; p.res_rows = 0;
  p.M = B * OH * OW; p.K = KH * KW * 4; p.act = act;
  return launch_gemm_x6rd_c4(p, w6, cout_pad, (hipStream_t)stream);
}

extern "C" int aot_conv2d_bf16_f32(const float* in, const void* wq, int cout_pad, const float* bias, const float* res, float* out,
                                   int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad,
                                   int dil, int lda, int ldc, int ldr, int res_rows, int act, int ksplit, float* scratch, void* stream) {
  if (!in || !wq || !out || ksplit < 1 || ksplit > 64 || (ksplit > 1 && !scratch)) return AOT_ERR_BADARG;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || OH <= 0 || OW <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return AOT_ERR_BADARG;
  if ((lda & 3) || lda < Cin || ldc < Cout) return AOT_ERR_BADARG;
  if (res && (ldr < Cout || res_rows < 0)) return AOT_ERR_BADARG;
  if ((long)B * OH * OW > 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  ConvParams p;
  p.in = in; p.w = nullptr; p.wt = nullptr; p.bias = bias; p.res = res; p.out = out;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.lda = lda; p.ldb = 0; p.ldwt = 0; p.ldc = ldc; p.ldr = ldr; p.res_rows = res_rows;
  p.M = B * OH * OW; p.K = KH * KW * Cin; p.act = act;
  return launch_gemm_x6(p, wq, cout_pad, 64, (hipStream_t)stream, 1, ksplit, scratch);
}
