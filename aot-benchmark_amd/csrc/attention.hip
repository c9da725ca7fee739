// Long-term / self attention over the memory bank, flash style, exact fp32 on the matrix cores.
//
// One 64-lane wave owns a tile of 32 queries of one head and streams the bank in tiles of 32 keys:
//
//   S^T[key, q]  = K_tile (32 x d) . Q^T (d x 32)      16 x v_mfma_f32_32x32x2_f32  (d = 32)
//   online softmax per query column (the query index is the LANE in the C/D layout, so the running
//   max m and the rescale factor are lane-local; the 32 keys of a column live in 16 registers of lanes
//   q and q+32 -> one cross-half exchange per tile for the max, none for the sum)
//   O^T[dv, q]  += V_tile^T (d x 32) . P^T (32 x 32)    16 x v_mfma_f32_32x32x2_f32
//
// Both products keep the QUERY on the B side, so P^T is consumed straight from the registers the
// first product left it in: the contraction index of the second product is simply enumerated in the
// order the C/D layout holds the keys (key(s, hi) = (s&3) + 8*(s>>2) + 4*hi) and V rows are fetched in
// that order.  No LDS, no P shuffles.  K/V are read as full 128-byte head slices of the token-major
// bank ([T, H*32] rows, coalesced); the bank (<= 48 MB/layer) lives in L2 / Infinity Cache across the
// 53 query tiles.
//
// Key-range parallelism comes in two levels.  The four waves of a workgroup take the four quarters of the workgroup's
// key range for the SAME (query tile, head) and merge their (O, m, l) through LDS at the end -- free of HBM traffic.
// nsplit > 1 additionally cuts the bank into nsplit contiguous ranges handled by different workgroups (only needed to
// fill the 1024 SIMDs for long banks), each writing an un-normalised partial (O, m, l) that attn_merge_kernel combines.
// With the in-workgroup level 4x fewer partial slabs exist than with a purely grid-level split (round 1: 12 slabs at
// M = 14, 3.3x the algorithmic bytes; now 3), and self-attention / short banks need no slab at all.
#include "common.h"
#include <type_traits>

struct AttnParams {
  const float* q;
  const float* k;
  const float* v;
  float* out;
  float* part;  // [nsplit][Nq][H*32] O partials, then [nsplit][Nq][H][2] (m, l)
  const int* T_dev;
  const float* gate;  // optional [Nq, ldg]: out *= gate (GatedPropagation, attention.py:707)
  int Nq, T, H, ldq, ldk, ldv, ldo, nsplit;
  int ldg, C;         // C = output width (H*32 for the multi-head form, dv for the gated form)
  int B;              // lanes (object groups / clips): lane b owns query rows [b*Nq, (b+1)*Nq) of q / out / gate / part
  long kv_brows;      // ... and key/value rows [b*kv_brows, b*kv_brows + T) of k / v (one bank per lane)
  float scale_div;
};

// V (and, in the gated kernels, wide-V) rows are fetched through a buffer descriptor: per-lane offset loop invariant,
// row part a wave-uniform scalar, rows >= T read as 0 by the hardware bounds check.  The descriptor is based at the
// first row of the wave's own key range (64-bit pointer arithmetic) and covers [row0, min(T, row1 + 64)) only, so its
// 32-bit byte count / offsets never see the size of the whole bank (a DeAOT bank of 313 memorised 480p frames is > 2 GB).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t v_descriptor(const float* v, long lane_row0, int row0, int row1, int T,
                                                                int ldv) {
  const int rows = max(0, min(T, row1 + 64) - row0);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(v + (lane_row0 + row0) * ldv), 0, rows * ldv * 4, 0x00020000);
}

// Softmax in the log2 domain.  With L = log2(e) and a running reference mL = fl(max_score * L), every weight is
//   p = 2^(fl(s * L - mL))          one v_fma_f32 + one v_exp_f32 per score
// The value of mL only has to be the SAME number wherever it is used (weights, row sums, rescale factors 2^(mL_old -
// mL_new), and the split merge, which receives mL itself), so its own rounding cancels in the normalisation; what is
// left is the rounding of the fused multiply-add, a relative error of |t| * 2^-24 * ln2 on a weight of size 2^t, i.e.
// <= 3e-8 absolute on any weight -- the level of one fp32 rounding of the largest weight.  (The accurate libm expf
// costs ~20 VALU instructions per score; 16 scores per lane and tile made that as expensive as the tile's 32 MFMAs.)
// Masked scores (-inf) and the initial mL = -inf give 2^-inf = 0 exactly; no clamps needed.
#define AOT_LOG2E 1.44269502162933349609375f
// Measured variants of the softmax step (tools/dev/mb_attn.py on MI355X, M = 14 / 8): plain fmaxf / scalar fp32 364 / 220 us;
// v_max3 359 / 219; v_max3 + packed fp32 350 / 216 (what is built); unrolled by two to drop the score-tile copy 369-390 /
// 223-231 (the larger loop body costs more than the nine v_mov it saves); row sums through v_pk_add_f32 instead of the scalar
// adds hipcc emits 362.6 vs 354.2 at M = 14 and -1.2 % on the whole frame (round 3): rejected and removed.
#ifndef AOT_ATT_VPM
#define AOT_ATT_VPM 4
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
// max of three in ONE instruction (same NaN rule as fmaxf: a NaN operand is ignored)
// packed fp32: two values per lane and instruction (hipcc scalarises most <2 x float> arithmetic next to MFMA operands)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float exp2_w(float s, float mL) {
  return __builtin_amdgcn_exp2f(fmaf(s, AOT_LOG2E, -mL));
}

// ---------------------------------------------------------------------------------------------------------
// Software pipelining: the score MFMAs of key tile i+1 are issued in the same instruction stream as the softmax VALU
// work of tile i (they are independent), so a wave keeps the matrix pipe fed while it exponentiates, instead of relying
// on other waves being in a different phase (+10-12 % over the plain tile-by-tile order at every bank size).  Costs one
// more 16-register score tile (110 VGPRs, 4 waves per SIMD) and one wasted score tile at the end of a wave's key range.
// Variants measured and dropped: register double-buffering of K/V (162 VGPRs, 3 waves: -4 %); two
// query tiles per wave (NQ = 2, -DAOT_ATTN_NQ=2: halves the K/V fetch per MFMA but needs 168-182 VGPRs, 2-3 waves: 379 vs
// 370 us at M = 14, 123 vs 121 us at M = 4).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 4) attn_fwd_d32_pipe_kernel(const AttnParams p) {
  // 4 waves per workgroup (one per SIMD of the CU), 4 workgroups per CU -> 4 waves per SIMD
  __shared__ float red[4][18][64];      // per wave: o[16], m, l  (18 KB)
  const int h = blockIdx.x, split = blockIdx.z, bz = blockIdx.y;
  const int ntq = (p.Nq + 31) >> 5;
  const int b = bz / ntq, qt = bz - b * ntq;
  const int lane = threadIdx.x & 63, j = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int T = p.T_dev ? *p.T_dev : p.T;
  const int ntile = (T + 31) >> 5;
  const int tps = (ntile + p.nsplit - 1) / p.nsplit;        // key tiles of this workgroup's range
  const int tpw = (tps + 3) >> 2;                           // ... and of each wave's quarter
  const int s1 = min(T, (split + 1) * tps * 32);            // end of the workgroup's range
  const int t0 = min(s1, (split * tps + wave * tpw) * 32);
  const int t1 = min(s1, t0 + tpw * 32);
  const long qrow0 = (long)b * p.Nq, krow0 = (long)b * p.kv_brows;

  float qf[16];
  {
    const int qrow = min(qt * 32 + j, p.Nq - 1);
    const float4* src = reinterpret_cast<const float4*>(p.q + (qrow0 + qrow) * p.ldq + h * 32 + hi * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 t = src[i];
      qf[4 * i + 0] = t.x / p.scale_div;    // the reference divides (attention.py:82), so do we
      qf[4 * i + 1] = t.y / p.scale_div;
      qf[4 * i + 2] = t.z / p.scale_div;
      qf[4 * i + 3] = t.w / p.scale_div;
    }
  }
  // (K uses plain 16-byte global loads: the raw_buffer_load_b64/b96/b128 builtins of ROCm 7.2's hipcc lower to a
  //  single buffer_load_dword -- verified in the ISA -- so only the 4-byte form is usable for V.)
  const __amdgpu_buffer_rsrc_t rv = v_descriptor(p.v, krow0, t0, t1, T, p.ldv);
  const int vvoff = (4 * hi * p.ldv + h * 32 + j) * 4;    // V: lane = channel j; rows 4*hi + (s&3) + 8*(s>>2)
  const int ldv4 = p.ldv * 4;
  const float* kptr = p.k + krow0 * p.ldk + h * 32 + hi * 16;

  float m = -INFINITY, l = 0.f;   // m: running max score times log2(e)
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;

  auto load_k = [&](float (&kf)[16], int kt) {
    const float4* src = reinterpret_cast<const float4*>(kptr + (long)min(kt + j, T - 1) * p.ldk);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 t = src[i];
      kf[4 * i] = t.x; kf[4 * i + 1] = t.y; kf[4 * i + 2] = t.z; kf[4 * i + 3] = t.w;
    }
  };
  auto load_v = [&](float (&vf)[16], int kt) {      // kt relative to the descriptor's first row t0
#pragma unroll
    for (int s = 0; s < 16; ++s)
      vf[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, vvoff, (kt - t0 + (s & 3) + 8 * (s >> 2)) * ldv4, 0));
  };
  auto qk = [&](const float (&kf)[16], f32x16& sc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], sc, 0, 0, 0);
  };

  float ka[16], va[16];
  f32x16 sa, sb;      // score tiles: the step of tile i reads one and fills the other with tile i+1
  if (t0 < t1) {
    load_k(ka, t0);
    load_v(va, t0);
    qk(ka, sa);
    load_k(ka, t0 + 32);
  }
  // One straight-line step (no branches, so the scheduler can interleave the next tile's score MFMAs with this tile's
  // softmax VALU work).  TAIL = the last, possibly partial tile of the range: keys >= t1 are masked to -inf.
  // VALU budget: on gfx950 a v_mfma_f32_32x32x2_f32 runs on the fp32 vector ALUs, so VALU work does NOT hide under it
  // (tools/dev/mfma_filler.hip: +5-6 cycles per VALU instruction beside the chain, +11 per v_exp_f32, at any occupancy) --
  // every instruction of the softmax comes straight out of the MFMA rate.  Hence: v_max3_f32 for the tile maximum (inline
  // asm: fmaxf() adds a canonicalising v_max per MFMA output) and packed fp32 (v_pk_fma / v_pk_add / v_pk_mul: two values
  // per lane and instruction) for the exponent arguments, the row sums and the rescale: 103 -> 80 VALU instructions per tile.
  auto step = [&](int kt, f32x16& sc, f32x16& scn, auto tail) {
    constexpr bool TAIL = decltype(tail)::value;
    // scores of the NEXT tile (past the range end: clamped rows, result unused) -- independent of everything below
    qk(ka, scn);
    if (TAIL) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt + mfma32_row(r, hi) >= t1) sc[r] = -INFINITY;
    }
    float x = max3f(max3f(max3f(sc[0], sc[1], sc[2]), max3f(sc[3], sc[4], sc[5]), max3f(sc[6], sc[7], sc[8])),
                    max3f(sc[9], sc[10], sc[11]), max3f(sc[12], sc[13], max3f(sc[14], sc[15], sc[15])));
    const float mnew = fmaxf(m, fmaxf(x, __shfl_xor(x, 32)) * AOT_LOG2E);
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);   // 1 when the max did not move; 0 on the first tile
    m = mnew;
    l *= alpha;
    float pf[16];
    {
      const f32x2 al2 = {alpha, alpha};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t = pk_mul(f32x2{o[r], o[r + 1]}, al2);
        o[r] = t[0];
        o[r + 1] = t[1];
      }
    }
    const f32x2 L2 = {AOT_LOG2E, AOT_LOG2E}, nm2 = {-m, -m};
    f32x2 ps = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 s2 = {sc[r], sc[r + 1]};
      const f32x2 t2 = pk_fma(s2, L2, nm2);      // fl(s * log2(e) - mL) per element, as exp2_w()
      pf[r] = __builtin_amdgcn_exp2f(t2[0]);
      pf[r + 1] = __builtin_amdgcn_exp2f(t2[1]);
      const f32x2 p2 = {pf[r], pf[r + 1]};
      ps += p2;
    }
    l += ps[0] + ps[1];
    load_k(ka, kt + 64);     // K registers were consumed by qk() above
#pragma unroll
    for (int s = 0; s < 16; ++s) o = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s], pf[s], o, 0, 0, 0);
    load_v(va, kt + 32);     // rows past the descriptor read as 0
    // issue order: 16 x (1 score MFMA, a few softmax VALU), then the rest as the scheduler likes
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, AOT_ATT_VPM, 0);
    }
  };
  int kt = t0;
  for (; kt + 64 < t1; kt += 32) {
    step(kt, sa, sb, std::false_type{});
    sa = sb;
  }
  if (kt + 32 < t1) {          // two tiles left
    step(kt, sa, sb, std::false_type{});
    step(kt + 32, sb, sa, std::true_type{});
  } else if (kt < t1) {        // one
    step(kt, sa, sb, std::true_type{});
  }

  // ---- merge of the four key quarters through LDS (fixed wave order: deterministic) ----
  {
    const float lt = l + __shfl_xor(l, 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = o[r];
    red[wave][16][lane] = m;      // log2 domain; -inf if this wave saw no key
    red[wave][17][lane] = lt;
  }
  __syncthreads();
  float mm = fmaxf(fmaxf(red[0][16][lane], red[1][16][lane]), fmaxf(red[2][16][lane], red[3][16][lane]));
  float f[4], lsum = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < 4; ++w2) {
    const float mw = red[w2][16][lane];
    f[w2] = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - mm);
    lsum += f[w2] * red[w2][17][lane];
  }
  // wave w finishes accumulator registers 4w..4w+3 = channels 8w + 4hi + (0..3): one float4 per lane
  float4 acc;
  {
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * wave + i;
      t[i] = merge4_scalar(f, red[0][r][lane], red[1][r][lane], red[2][r][lane], red[3][r][lane]);      // (scalar on purpose: common.h)
    }
    acc = make_float4(t[0], t[1], t[2], t[3]);
  }
  const int qi = qt * 32 + j;
  if (qi >= p.Nq) return;
  const long grow = qrow0 + qi;          // row in q / out / gate / part
  const int c = h * 32 + 8 * wave + 4 * hi;
  if (p.nsplit == 1) {
    const float inv = 1.f / lsum;
    // (each product through scalar_fp32: paired, hipcc wrote the result over the register that holds `inv` -- an in-place packed
    // multiply whose high half reads the overwritten low half; common.h)
    acc.x = scalar_fp32(acc.x * inv); acc.y = scalar_fp32(acc.y * inv); acc.z = scalar_fp32(acc.z * inv); acc.w = scalar_fp32(acc.w * inv);
    if (p.gate) {
      const float4 u = *reinterpret_cast<const float4*>(p.gate + grow * p.ldg + c);
      acc.x *= u.x; acc.y *= u.y; acc.z *= u.z; acc.w *= u.w;
    }
    *reinterpret_cast<float4*>(p.out + grow * p.ldo + c) = acc;
  } else {
    const int C = p.H * 32;
    const long rows = (long)p.B * p.Nq;
    *reinterpret_cast<float4*>(p.part + ((long)split * rows + grow) * C + c) = acc;
    if (wave == 0 && hi == 0) {
      float* ml = p.part + (long)p.nsplit * rows * C + (((long)split * rows + grow) * p.H + h) * 2;
      ml[0] = mm;      // log2 domain; -inf if this split saw no key
      ml[1] = lsum;
    }
  }
}

// merge of the nsplit partials: one thread per (query, 4 channels).  Stats (m, l) are stored per `group` of
// p.C / p.H channels (one per head in the multi-head form, one per V chunk in the gated form).
__global__ void __launch_bounds__(256) attn_merge_kernel(const AttnParams p) {
  const int C = p.C;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)p.Nq * (C / 4);
  if (idx >= total) return;
  const int c4 = (int)(idx % (C / 4));
  const int qi = (int)(idx / (C / 4));
  const int h = (c4 * 4) / (C / p.H);
  const float* mlb = p.part + (long)p.nsplit * p.Nq * C;
  float mmax = -INFINITY;
  for (int s = 0; s < p.nsplit; ++s) mmax = fmaxf(mmax, mlb[(((long)s * p.Nq + qi) * p.H + h) * 2]);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float lsum = 0.f;
  for (int s = 0; s < p.nsplit; ++s) {
    const float* ml = mlb + (((long)s * p.Nq + qi) * p.H + h) * 2;
    const float ms = ml[0];
    if (ms == -INFINITY) continue;
    const float wgt = __builtin_amdgcn_exp2f(ms - mmax);   // stats are kept in the log2 domain
    lsum += wgt * ml[1];
    const float4 t = *reinterpret_cast<const float4*>(p.part + ((long)s * p.Nq + qi) * C + c4 * 4);
    acc.x += wgt * t.x; acc.y += wgt * t.y; acc.z += wgt * t.z; acc.w += wgt * t.w;
  }
  const float inv = 1.f / lsum;
  float4 r = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  if (p.gate) {
    const float4 u = *reinterpret_cast<const float4*>(p.gate + (long)qi * p.ldg + c4 * 4);
    r.x *= u.x; r.y *= u.y; r.z *= u.z; r.w *= u.w;
  }
  *reinterpret_cast<float4*>(p.out + (long)qi * p.ldo + c4 * 4) = r;
}

// ---------------------------------------------------------------------------------------------------------
// Gated-propagation form (DeAOT, attention.py:672-707): ONE query/key head of width DQK = 128 and a value of
// width NCH * 32*NDV (1024 = [V | ID_V]).  A wave owns 32 queries and one chunk of 32*NDV value channels (NDV = 8:
// 128 accumulator registers, one wave per SIMD).
// ---------------------------------------------------------------------------------------------------------
// General form (any number of chunks; used when dv != 1024): every wave recomputes the score tile of its queries
// (DQK/2 MFMAs, a third of its work at NDV = 8).  Software-pipelined like attn_fwd_d32_pipe_kernel: the score MFMAs of
// key tile i+1 share an instruction stream with the softmax of tile i, K needs a single register set (its next load is
// issued as soon as the score MFMAs have consumed it and lands under the 16*NDV value MFMAs), and the value chunks are
// fetched TWO chunks ahead through three rotating register sets (one wave per SIMD: nothing else hides that latency).
template <int DQK, int NDV>
__global__ void __launch_bounds__(64) attn_fwd_wide_pipe_kernel(const AttnParams p) {
  constexpr int HK = DQK / 2;
  const int ch = blockIdx.x, split = blockIdx.y;
  const int ntq = (p.Nq + 31) >> 5;
  const int b = blockIdx.z / ntq, qt = blockIdx.z - b * ntq;
  const int lane = threadIdx.x, j = lane & 31, hi = lane >> 5;
  const int T = p.T_dev ? *p.T_dev : p.T;
  const int ntile = (T + 31) >> 5;
  const int tps = (ntile + p.nsplit - 1) / p.nsplit;
  const int t0 = min(T, split * tps * 32);
  const int t1 = min(T, t0 + tps * 32);
  const int qrow = min(qt * 32 + j, p.Nq - 1);
  const long qrow0 = (long)b * p.Nq, krow0 = (long)b * p.kv_brows;

  float qf[HK];
  {
    const float4* src = reinterpret_cast<const float4*>(p.q + (qrow0 + qrow) * p.ldq + hi * HK);
#pragma unroll
    for (int i = 0; i < HK / 4; ++i) {
      const float4 t = src[i];
      qf[4 * i] = t.x / p.scale_div; qf[4 * i + 1] = t.y / p.scale_div;
      qf[4 * i + 2] = t.z / p.scale_div; qf[4 * i + 3] = t.w / p.scale_div;
    }
  }
  const __amdgpu_buffer_rsrc_t rv = v_descriptor(p.v, krow0, t0, t1, T, p.ldv);
  const int vvoff = (4 * hi * p.ldv + ch * 32 * NDV + j) * 4;
  const int ldv4 = p.ldv * 4;
  const float* kptr = p.k + krow0 * p.ldk + hi * HK;

  float m = -INFINITY, l = 0.f;   // m in the log2 domain
  f32x16 o[NDV];
#pragma unroll
  for (int d = 0; d < NDV; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;

  auto load_k = [&](float (&kf)[HK], int kt) {
    const float4* src = reinterpret_cast<const float4*>(kptr + (long)min(kt + j, T - 1) * p.ldk);
#pragma unroll
    for (int i = 0; i < HK / 4; ++i) {
      const float4 t = src[i];
      kf[4 * i] = t.x; kf[4 * i + 1] = t.y; kf[4 * i + 2] = t.z; kf[4 * i + 3] = t.w;
    }
  };
  auto load_v = [&](float (&vf)[16], int kt, int d) {
#pragma unroll
    for (int s = 0; s < 16; ++s)
      vf[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, vvoff + d * 128, (kt - t0 + (s & 3) + 8 * (s >> 2)) * ldv4, 0));
  };
  auto qk = [&](const float (&kf)[HK]) {
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < HK; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], sc, 0, 0, 0);
    return sc;
  };

  float ka[HK], vb[3][16];
  f32x16 sc;
  if (t0 < t1) {
    load_k(ka, t0);
    sc = qk(ka);
    load_k(ka, t0 + 32);
  }
  auto step = [&](int kt, auto tail) {
    constexpr bool TAIL = decltype(tail)::value;
    load_v(vb[0], kt, 0);
    load_v(vb[1], kt, 1);
    f32x16 scn = qk(ka);     // next tile's scores: independent of the softmax below
    if (TAIL) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt + mfma32_row(r, hi) >= t1) sc[r] = -INFINITY;
    }
    float x = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
#pragma unroll
    for (int r = 4; r < 16; r += 4) x = fmaxf(x, fmaxf(fmaxf(sc[r], sc[r + 1]), fmaxf(sc[r + 2], sc[r + 3])));
    const float mnew = fmaxf(m, fmaxf(x, __shfl_xor(x, 32)) * AOT_LOG2E);
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);
    const bool moved = mnew > m;
    m = mnew;
    l *= alpha;
    float pf[16];
    float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      pf[r] = exp2_w(sc[r], m);
      pf[r + 1] = exp2_w(sc[r + 1], m);
      ps0 += pf[r];
      ps1 += pf[r + 1];
    }
    l += ps0 + ps1;
    load_k(ka, kt + 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) {     // issue order: score MFMAs with the softmax VALU work threaded between them
      __builtin_amdgcn_sched_group_barrier(0x008, HK / 16, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
    }
    if (__any(moved)) {   // the 16*NDV accumulator registers are only rescaled when some query's max moved
#pragma unroll
      for (int d = 0; d < NDV; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
#pragma unroll
    for (int d = 0; d < NDV; ++d) {
      if (d + 2 < NDV) load_v(vb[(d + 2) % 3], kt, d + 2);
#pragma unroll
      for (int s = 0; s < 16; ++s) o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[d % 3][s], pf[s], o[d], 0, 0, 0);
    }
    sc = scn;
  };
  int kt = t0;
  for (; kt + 32 < t1; kt += 32) step(kt, std::false_type{});
  if (kt < t1) step(kt, std::true_type{});

  l += __shfl_xor(l, 32);
  if (qt * 32 + j >= p.Nq) return;
  const long qi = qrow0 + qt * 32 + j;          // row in q / out / gate / part
  const long prow = (long)p.B * p.Nq;          // rows per partial slab
  const int cbase = ch * 32 * NDV + 4 * hi;
  if (p.nsplit == 1) {
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < NDV; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 t = make_float4(o[d][4 * g] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
        const int c = cbase + d * 32 + 8 * g;
        if (p.gate) {
          const float4 u = *reinterpret_cast<const float4*>(p.gate + qi * p.ldg + c);
          t.x *= u.x; t.y *= u.y; t.z *= u.z; t.w *= u.w;
        }
        *reinterpret_cast<float4*>(p.out + qi * p.ldo + c) = t;
      }
  } else {
    float* dst = p.part + ((long)split * prow + qi) * p.C;
#pragma unroll
    for (int d = 0; d < NDV; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dst + cbase + d * 32 + 8 * g) =
            make_float4(o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]);
    if (hi == 0) {
      float* ml = p.part + (long)p.nsplit * prow * p.C + (((long)split * prow + qi) * p.H + ch) * 2;
      ml[0] = m;
      ml[1] = l;
    }
  }
}

// Cooperative form of the gated-propagation kernel for dv = 4 x 256: the four waves of a workgroup own the four value
// chunks of the SAME 32 queries and share the score tile instead of recomputing it -- wave w contracts channels
// [32w, 32w+32) of q.k (DQK/8 MFMAs instead of DQK/2), the four partial tiles meet in LDS (summed in wave order by every
// wave, so all four hold bit-identical scores) and each wave runs the softmax and its own 16*NDV value MFMAs.  144 instead
// of 192 MFMAs per key tile and wave, q/k fragments of 16 registers instead of 64.  One barrier per key tile; the score
// partials of tile i+1 are produced (software-pipelined, as above) while tile i is being exponentiated.
// Two waves per SIMD (round 3): the registers are capped at 256 (launch bound 2) and V is fetched ONE chunk ahead through two
// rotating register sets instead of two ahead through three.  With one wave per SIMD every stall -- the per-tile barrier, a late V
// row -- idled the matrix pipe (PMC: MFMA busy 52 %, waves parked 35 %); the second wave fills those holes although the capped
// allocation makes hipcc park ~300 values per key tile in spare AGPRs: R50-DeAOTL 262.6 -> 306.4 fps with three clips per GPU,
// 212 -> 219 one clip at a time (profiles/r03f_gated_occ2.txt).
template <int NDV>
__global__ void __launch_bounds__(256, 2) attn_fwd_wide_coop_kernel(const AttnParams p) {
  constexpr int NVB = 2;       // rotating V register sets
  const int ntq = (p.Nq + 31) >> 5;
  // XCD-major block order (round 6, as attn_x6_wide64p_kernel): block id -> XCD id % 8, and each XCD takes a contiguous run of the
  // (lane, key range, query tile) triples in key-range-major order, so that its L2 streams one or two key ranges.  With the key
  // range in blockIdx.x every XCD pulled the whole bank: 621.6 MB of fabric traffic per launch, 9x the algorithmic bytes
  // (profiles/r04_gated_attn_traffic.json).
  int split, b, qt;
  {
    const int total = p.B * p.nsplit * ntq, per = (total + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = xcd * per + slot;
    if (slot >= per || pair >= total) return;
    qt = pair % ntq;
    const int bs = pair / ntq;
    split = bs % p.nsplit;
    b = bs / p.nsplit;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, hi = lane >> 5;
  const int ch = wave;
  const int T = p.T_dev ? *p.T_dev : p.T;
  const int ntile = (T + 31) >> 5;
  const int tps = (ntile + p.nsplit - 1) / p.nsplit;
  const int t0 = min(T, split * tps * 32);
  const int t1 = min(T, t0 + tps * 32);
  const int qrow = min(qt * 32 + j, p.Nq - 1);
  const long qrow0 = (long)b * p.Nq, krow0 = (long)b * p.kv_brows;
  __shared__ float part[2][4][16][64];     // [buffer][wave][score register][lane]: 32 KB, conflict-free b32 accesses

  float qf[16];
  {
    const float4* src = reinterpret_cast<const float4*>(p.q + (qrow0 + qrow) * p.ldq + wave * 32 + hi * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 t = src[i];
      qf[4 * i] = t.x / p.scale_div; qf[4 * i + 1] = t.y / p.scale_div;
      qf[4 * i + 2] = t.z / p.scale_div; qf[4 * i + 3] = t.w / p.scale_div;
    }
  }
  const __amdgpu_buffer_rsrc_t rv = v_descriptor(p.v, krow0, t0, t1, T, p.ldv);
  const int vvoff = (4 * hi * p.ldv + ch * 32 * NDV + j) * 4;
  const int ldv4 = p.ldv * 4;
  const float* kptr = p.k + krow0 * p.ldk + wave * 32 + hi * 16;

  float m = -INFINITY, l = 0.f;   // m in the log2 domain
  f32x16 o[NDV];
#pragma unroll
  for (int d = 0; d < NDV; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;

  auto load_k = [&](float (&kf)[16], int kt) {
    const float4* src = reinterpret_cast<const float4*>(kptr + (long)min(kt + j, T - 1) * p.ldk);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 t = src[i];
      kf[4 * i] = t.x; kf[4 * i + 1] = t.y; kf[4 * i + 2] = t.z; kf[4 * i + 3] = t.w;
    }
  };
  auto load_v = [&](float (&vf)[16], int kt, int d) {
#pragma unroll
    for (int s = 0; s < 16; ++s)
      vf[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, vvoff + d * 128, (kt - t0 + (s & 3) + 8 * (s >> 2)) * ldv4, 0));
  };
  auto qk_part = [&](const float (&kf)[16], int buf) {     // this wave's 32-channel share of the score tile -> LDS
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], sc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) part[buf][wave][r][lane] = sc[r];
  };

  float ka[16], vb[NVB][16];
  if (t0 < t1) {
    load_k(ka, t0);
    qk_part(ka, 0);
    load_k(ka, t0 + 32);
  }
  __syncthreads();
  int it = 0;
  auto step = [&](int kt, auto tail) {
    constexpr bool TAIL = decltype(tail)::value;
    const int buf = it & 1;
    load_v(vb[0], kt, 0);
    if (NVB > 2) load_v(vb[1], kt, 1);
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {   // fixed wave order: every wave of the workgroup gets the same bits (two rows per v_pk_add)
      const f32x2 p0 = {part[buf][0][r][lane], part[buf][0][r + 1][lane]}, p1 = {part[buf][1][r][lane], part[buf][1][r + 1][lane]};
      const f32x2 p2 = {part[buf][2][r][lane], part[buf][2][r + 1][lane]}, p3 = {part[buf][3][r][lane], part[buf][3][r + 1][lane]};
      const f32x2 t = pk_add(pk_add(pk_add(p0, p1), p2), p3);
      sc[r] = t[0];
      sc[r + 1] = t[1];
    }
    qk_part(ka, buf ^ 1);            // next tile's partial scores: independent of the softmax below
    if (TAIL) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt + mfma32_row(r, hi) >= t1) sc[r] = -INFINITY;
    }
    // (same VALU diet as the d = 32 kernel: v_max3, packed fp32 -- fp32 MFMAs do not hide VALU work)
    const float x = max3f(max3f(max3f(sc[0], sc[1], sc[2]), max3f(sc[3], sc[4], sc[5]), max3f(sc[6], sc[7], sc[8])),
                          max3f(sc[9], sc[10], sc[11]), max3f(sc[12], sc[13], max3f(sc[14], sc[15], sc[15])));
    const float mnew = fmaxf(m, fmaxf(x, __shfl_xor(x, 32)) * AOT_LOG2E);
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);
    const bool moved = mnew > m;
    m = mnew;
    l *= alpha;
    float pf[16];
    {
      const f32x2 L2 = {AOT_LOG2E, AOT_LOG2E}, nm2 = {-m, -m};
      f32x2 ps = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t2 = pk_fma(f32x2{sc[r], sc[r + 1]}, L2, nm2);
        pf[r] = __builtin_amdgcn_exp2f(t2[0]);
        pf[r + 1] = __builtin_amdgcn_exp2f(t2[1]);
        ps += f32x2{pf[r], pf[r + 1]};
      }
      l += ps[0] + ps[1];
    }
    load_k(ka, kt + 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
    }
    if (__any(moved)) {
      // explicit AGPR reads / writes: written as plain C++ the (rarely taken) rescale made hipcc copy the 128 accumulators
      // to VGPRs and back around the branch on EVERY key tile (~270 v_accvgpr moves per step); +2 % on R50-DeAOTL (round 3)
      const f32x2 al2 = {alpha, alpha};
#pragma unroll
      for (int d = 0; d < NDV; ++d)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float a0, a1;
          asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(a0), "=v"(a1) : "a"(o[d][r]), "a"(o[d][r + 1]));
          const f32x2 t = pk_mul(f32x2{a0, a1}, al2);
          float w0, w1;
          asm volatile("v_accvgpr_write_b32 %0, %2\n\tv_accvgpr_write_b32 %1, %3" : "=a"(w0), "=a"(w1) : "v"(t[0]), "v"(t[1]));
          o[d][r] = w0;
          o[d][r + 1] = w1;
        }
    }
#pragma unroll
    for (int d = 0; d < NDV; ++d) {
      if (d + NVB - 1 < NDV) load_v(vb[(d + NVB - 1) % NVB], kt, d + NVB - 1);
#pragma unroll
      for (int s = 0; s < 16; ++s) o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[d % NVB][s], pf[s], o[d], 0, 0, 0);
    }
    ++it;
    __syncthreads();     // next tile's partials visible; this tile's buffer free for the tile after next
  };
  int kt = t0;
  for (; kt + 32 < t1; kt += 32) step(kt, std::false_type{});
  if (kt < t1) step(kt, std::true_type{});

  l += __shfl_xor(l, 32);
  if (qt * 32 + j >= p.Nq) return;
  const long qi = qrow0 + qt * 32 + j;          // row in q / out / gate / part
  const long prow = (long)p.B * p.Nq;          // rows per partial slab
  const int cbase = ch * 32 * NDV + 4 * hi;
  if (p.nsplit == 1) {
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < NDV; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 t = make_float4(o[d][4 * g] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
        const int c = cbase + d * 32 + 8 * g;
        if (p.gate) {
          const float4 u = *reinterpret_cast<const float4*>(p.gate + qi * p.ldg + c);
          t.x *= u.x; t.y *= u.y; t.z *= u.z; t.w *= u.w;
        }
        *reinterpret_cast<float4*>(p.out + qi * p.ldo + c) = t;
      }
  } else {
    float* dst = p.part + ((long)split * prow + qi) * p.C;
#pragma unroll
    for (int d = 0; d < NDV; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dst + cbase + d * 32 + 8 * g) =
            make_float4(o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]);
    if (hi == 0) {
      float* ml = p.part + (long)p.nsplit * prow * p.C + (((long)split * prow + qi) * p.H + ch) * 2;
      ml[0] = m;
      ml[1] = l;
    }
  }
}

static int fill_params(AttnParams& p, const float* q, const float* k, const float* v, float* out, float* part, int B,
                       long kv_brows, int Nq, int T, const int* T_dev, int H, int ldq, int ldk, int ldv, int ldo,
                       float scale_div, int nsplit) {
  if (!q || !k || !v || !out || Nq <= 0 || T <= 0 || H <= 0 || B <= 0) return AOT_ERR_BADARG;
  if (B > 1 && kv_brows < T) return AOT_ERR_BADARG;
  // 32-bit byte offsets inside one workgroup's key range (the descriptor is re-based per range, see v_descriptor)
  if (((long)((T + 31) / 32 + nsplit - 1) / nsplit * 32 + 96) * ldv * 4 >= 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  if ((long)B * cdiv(Nq, 32) > 65535) return AOT_ERR_UNSUPPORTED;
  if ((ldq & 3) || (ldk & 3) || (ldo & 3) || ((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)out & 15))
    return AOT_ERR_BADARG;
  if (nsplit < 1) return AOT_ERR_BADARG;
  if (nsplit > 1 && !part) return AOT_ERR_BADARG;
  p.q = q; p.k = k; p.v = v; p.out = out; p.part = part; p.T_dev = T_dev; p.gate = nullptr; p.ldg = 0;
  p.Nq = Nq; p.T = T; p.H = H; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.nsplit = nsplit;
  p.B = B; p.kv_brows = kv_brows;
  p.C = H * 32;
  p.scale_div = scale_div;
  return AOT_OK;
}

extern "C" int aot_attn_f32(const float* q, const float* k, const float* v, float* out, float* part, int B,
                            long kv_brows, int Nq, int T, const int* T_dev, int H, int d, int ldq, int ldk, int ldv,
                            int ldo, float scale_div, int nsplit, void* stream) {
  if (d != 32) return AOT_ERR_UNSUPPORTED;
  AttnParams p;
  const int rc = fill_params(p, q, k, v, out, part, B, kv_brows, Nq, T, T_dev, H, ldq, ldk, ldv, ldo, scale_div, nsplit);
  if (rc) return rc;
  // Workgroup b of a grid runs on XCD b % 8 (observed dispatch rule), so with H = 8 heads in blockIdx.x every XCD serves one
  // head: its L2 sees that head's K / V slices only.  The grid has 53 x nsplit workgroups per head on 128 resident slots
  // per XCD; the ones that start in the second dispatch round re-stream their key range.  Split-major order (key range in
  // blockIdx.z, query tile in .y) makes that second round the tail of ONE key range instead of a slice of all of them.
  // Measured on the launch mix of a 70-frame clip (two PMC passes, profiles/r03_attn_traffic*.json): fabric traffic
  // 37.4 -> 20.3 MB per launch = 2.10x -> 1.14x the algorithmic 17.8 MB, at the same launch time (353 vs 354 us at M = 14).
  hipLaunchKernelGGL(attn_fwd_d32_pipe_kernel, dim3(H, B * cdiv(Nq, 32), nsplit), dim3(256), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_attn_merge_f32(const float* part, const float* gate, float* out, int Nq, int H, int C, int ldg,
                                  int ldo, int nsplit, void* stream) {
  if (!part || !out || Nq <= 0 || H <= 0 || C <= 0 || (C % H) || ((C / H) & 3) || nsplit < 2 || (ldo & 3)) return AOT_ERR_BADARG;
  if (gate && (ldg & 3)) return AOT_ERR_BADARG;
  AttnParams p = {};
  p.part = const_cast<float*>(part); p.out = out; p.Nq = Nq; p.H = H; p.C = C; p.ldo = ldo; p.nsplit = nsplit;
  p.gate = gate; p.ldg = ldg;
  const long total = (long)Nq * (C / 4);
  hipLaunchKernelGGL(attn_merge_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_gated_attn_f32(const float* q, const float* k, const float* v, const float* gate, float* out,
                                  float* part, int B, long kv_brows, int Nq, int T, const int* T_dev, int dqk, int dv,
                                  int ldq, int ldk, int ldv, int ldg, int ldo, float scale_div, int nsplit,
                                  void* stream) {
  if (dqk != 128 || dv <= 0 || (dv % 256)) return AOT_ERR_UNSUPPORTED;
  const int nch = dv / 256;
  AttnParams p;
  const int rc = fill_params(p, q, k, v, out, part, B, kv_brows, Nq, T, T_dev, nch, ldq, ldk, ldv, ldo, scale_div, nsplit);
  if (rc) return rc;
  if (gate && (ldg & 3)) return AOT_ERR_BADARG;
  p.C = dv;
  p.gate = (nsplit == 1) ? gate : nullptr;   // with splits the gate is applied by aot_attn_merge_f32
  p.ldg = ldg;
  if (nch == 4)     // dv = 1024 (every DeAOT config): the four chunk waves share one score tile
    hipLaunchKernelGGL((attn_fwd_wide_coop_kernel<8>), dim3(8 * cdiv((long)B * nsplit * cdiv(Nq, 32), 8)), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((attn_fwd_wide_pipe_kernel<128, 8>), dim3(nch, nsplit, B * cdiv(Nq, 32)), dim3(64), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}
