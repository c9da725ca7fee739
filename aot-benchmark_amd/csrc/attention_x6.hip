// Long-term / self attention of AOT on the bf16 matrix cores with fp32-equivalent arithmetic: the attention member of the
// bf16x6 kernel family (gemm_lds.hip holds the conv / linear members and the argument: every fp32 number is EXACTLY the sum of
// three bf16 numbers obtained by truncation; of the nine partial products the six of order <= 2 are kept, each exact in the
// fp32 accumulator of v_mfma_f32_32x32x16_bf16; what is dropped is <= 3 * 2^-24 of a product).  Same algorithm, grid, key
// split, partial-slab format and merge as attn_fwd_d32_pipe_kernel (attention.hip; reference attention.py:92-117); what
// differs is where the operands come from:
//
//   * the BANK is kept pre-split and TILE-MAJOR (aot_attn_pack_x6_f32, once per memorised frame; every later frame reads it):
//       kv[lane][row / 32][head][K: 3 planes x 2 sub-steps x 64 lanes x 8 | V: the same]  bf16, 12 KB per (32-row tile, head)
//     i.e. every MFMA operand fetch of a wave -- 64 lanes x 16 bytes -- is ONE contiguous KB (8 cache lines), laid out in the
//     lane order of the instruction.  K chunk (plane, c) holds for lane (row j, half hi) the dims 16 c + 8 hi .. + 8 of bank row
//     32 tile + j; V is stored TRANSPOSED: chunk (plane, c) holds for lane (dim j, half hi) the eight bank rows the score tile's
//     C/D layout keeps in that lane's registers 8 c .. 8 c + 7, so that P^T goes from the accumulator registers of the first
//     product straight into the B operand of the second (no LDS, no shuffles -- as in the fp32 kernel).  (A row-major packed
//     bank was measured first: each 16-byte fetch then touches 32 cache lines and the kernel is bound by the L1 tag rate --
//     no faster than the fp32 kernel.)
//   * Q is split once per wave (registers), P once per key tile (4 VALU per value + 3 v_perm per pair);
//   * a key tile is 24 MFMAs of 32 cycles instead of 32 of 64 -- and, unlike fp32 MFMAs, these do not occupy the vector
//     ALUs the softmax runs on.
#include "common.h"
#include <type_traits>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct AttnX6Params {
  const float* q;
  const unsigned short* kv;   // [B][cap_rows / 32][H][6144]
  float* out;
  float* part;          // [nsplit][B*Nq][H*32] O partials, then [nsplit][B*Nq][H][2] (m, l): attn_merge_kernel's format
  const int* T_dev;
  int Nq, T, H, ldq, ldo, nsplit, B;
  long cap_rows;        // rows of one lane's packed bank (a multiple of 32)
  float scale_div;
};

#define AOT_LOG2E 1.44269502162933349609375f

// two truncated bf16 (the upper halves of a and b) in one dword: [a.hi16 | b.hi16 << 16]
__device__ __forceinline__ unsigned pack_hi16(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// eight fp32 values -> their three bf16 planes (hi, mid, lo), element order preserved
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 (&out)[3]) {
  u32x4 w[3];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float r0 = x[2 * e], r1 = x[2 * e + 1];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      w[pl][e] = pack_hi16(r0, r1);
      if (pl < 2) {
        r0 -= __uint_as_float(__float_as_uint(r0) & 0xffff0000u);
        r1 -= __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) out[pl] = __builtin_bit_cast(bf16x8, w[pl]);
}

// acc += the six kept partial products of (a0 + a1 + a2) x (b0 + b1 + b2), smallest first
__device__ __forceinline__ void mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// Round 3 shipped this kernel with a warning that its instruction order was part of its correctness: some orders gave wrong O for
// 16 of the 32 queries of sporadic tiles, and the MFMA wait states were suspected.  Round 4 found the cause, and it is not in the
// loop (profiles/r04_hazard.txt): the (O, m, l) LDS merge at the end was compiled by hipcc's SLP vectoriser into an IN-PLACE packed
// add whose op_sel crosses the halves of the overwritten source (v_pk_add_f32 v[0:1], v[18:19], v[0:1] op_sel:[0,1] op_sel_hi:[1,0]),
// and gfx950 returns a wrong low half for lanes 48-63 of that instruction when other waves are active on the SIMD -- which waves
// are active when the merge runs is what the loop's instruction order changed.  Probes on the hardware confirmed hipcc's MFMA
// table on the way (XDL write -> VALU read 12 wait states, VALU write -> XDL SrcA/B 1, trans -> VALU 1, at 1 / 2 / 4 waves per
// SIMD).  The merge is now scalar (merge4_scalar, common.h), every kernel of the library is audited for the pattern
// (tests/test_host.py::test_no_inplace_crossed_packed_ops), and tools/dev/x6_hazard.hip keeps the deterministic reproducer.
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// Variants measured on MI355X and dropped (tools/dev/mb_attn_x6.py, profiles/r03n_attn_x6_variants.txt; N = 1674, 8 heads,
// bank of 4 / 14 frames: 80.8 / 243.2 us as built): sched_group_barrier interleaves of the MFMAs with the softmax / split VALU
// (82.1-83.2 / 246.8-249.4), both halves of P split before the value MFMAs (81.0 / 248.7), rescaling the accumulator only when
// some lane's maximum moved (a wave-uniform branch: 84.0-87.5 / 250-255).  hipcc's own order is the fastest.
// Round 4 (profiles/r04_pk_variants.txt; as built 81.5 / 245.3): fewer VALU instructions do not make it faster -- the residual
// subtractions of the splits as v_pk_add_f32 on register pairs (-16 per tile: 81.7 / 247.2; hipcc's post-RA pass unpacks a third
// of them again inside MFMA shadows, where packed fp32 cannot co-issue), the exponent's fma packed as well (81.9 / 248.8), two
// tiles per loop trip without the 16-register score copy (82.9 / 249.4), all three (82.5 / 251.3).  Same for the splits of the
// GEMM kernels (batch 3: 2413 -> 2424 us per frame set).  Nor do more waves: capped at 128 registers (four waves per SIMD instead
// of the three its 160 registers allow; the spills stay outside the loop) it is 15 % SLOWER (94.5 / 278 us,
// profiles/r04_x6_occupancy.txt).
// One wave = 32 queries x 1 head; the four waves of a workgroup take the four quarters of the workgroup's key range and merge
// through LDS; blockIdx = (head, lane * query tile, key split) exactly as attn_fwd_d32_pipe_kernel.
__global__ void __launch_bounds__(256, 2) attn_x6_d32_kernel(const AttnX6Params p) {
  __shared__ float red[4][18][64];
  const int h = blockIdx.x, split = blockIdx.z, bz = blockIdx.y;
  const int ntq = (p.Nq + 31) >> 5;
  const int b = bz / ntq, qt = bz - b * ntq;
  const int lane = threadIdx.x & 63, j = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int T = p.T_dev ? *p.T_dev : p.T;
  const int ntile = (T + 31) >> 5;
  const int tps = (ntile + p.nsplit - 1) / p.nsplit;
  const int tpw = (tps + 3) >> 2;
  const int s1 = min(T, (split + 1) * tps * 32);
  const int t0 = min(s1, (split * tps + wave * tpw) * 32);
  const int t1 = min(s1, t0 + tpw * 32);
  const long qrow0 = (long)b * p.Nq;
  const int C = p.H * 32;
  const long cap_tiles = p.cap_rows >> 5;

  // Q^T as the B operand of the score product: lane (query j, half hi) contracts dims 16 c + 8 hi .. + 8 in sub-step c
  bf16x8 qp[2][3];
  {
    const int qrow = min(qt * 32 + j, p.Nq - 1);
    const float* src = p.q + (qrow0 + qrow) * p.ldq + h * 32 + hi * 8;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float4 u0 = *reinterpret_cast<const float4*>(src + 16 * c), u1 = *reinterpret_cast<const float4*>(src + 16 * c + 4);
      float x[8] = {u0.x / p.scale_div, u0.y / p.scale_div, u0.z / p.scale_div, u0.w / p.scale_div,      // the reference divides
                    u1.x / p.scale_div, u1.y / p.scale_div, u1.z / p.scale_div, u1.w / p.scale_div};     // (attention.py:82)
      split3(x, qp[c]);
    }
  }
  // K: lane = key row j of the tile; V: lane = value channel j; chunk (plane, c) of either = 64 lanes x 16 bytes
  const unsigned short* kvbase = p.kv + ((long)b * cap_tiles * p.H + h) * 6144 + lane * 8;
  const long tile_stride = (long)p.H * 6144;

  float m = -INFINITY, l = 0.f;
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;

  auto load_k = [&](bf16x8 (&kf)[2][3], int kt) {
    const unsigned short* src = kvbase + min((long)(kt >> 5), cap_tiles - 1) * tile_stride;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int c = 0; c < 2; ++c) kf[c][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + (pl * 2 + c) * 512));
  };
  auto load_v = [&](bf16x8 (&vf)[2][3], int kt) {
    const unsigned short* src = kvbase + min((long)(kt >> 5), cap_tiles - 1) * tile_stride + 3072;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int c = 0; c < 2; ++c) vf[c][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + (pl * 2 + c) * 512));
  };
  auto qk = [&](const bf16x8 (&kf)[2][3], f32x16& sc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
    mfma6(kf[0], qp[0], sc);
    mfma6(kf[1], qp[1], sc);
  };

  bf16x8 ka[2][3], va[2][3];
  f32x16 sa, sb;
  if (t0 < t1) {
    load_k(ka, t0);
    load_v(va, t0);
    qk(ka, sa);
    load_k(ka, t0 + 32);
  }
  auto step = [&](int kt, f32x16& sc, f32x16& scn, auto tail) {
    constexpr bool TAIL = decltype(tail)::value;
    qk(ka, scn);      // scores of the NEXT tile (past the range end: clamped rows, result unused)
    if (TAIL) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt + mfma32_row(r, hi) >= t1) sc[r] = -INFINITY;
    }
    float x = max3f(max3f(max3f(sc[0], sc[1], sc[2]), max3f(sc[3], sc[4], sc[5]), max3f(sc[6], sc[7], sc[8])),
                    max3f(sc[9], sc[10], sc[11]), max3f(sc[12], sc[13], max3f(sc[14], sc[15], sc[15])));
    const float mnew = fmaxf(m, fmaxf(x, __shfl_xor(x, 32)) * AOT_LOG2E);
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);
    m = mnew;
    l *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;
    float pf[16], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pf[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], AOT_LOG2E, -m));      // as exp2_w() of attention.hip
      ps += pf[r];
    }
    l += ps;
    load_k(ka, kt + 64);
    // P^T as the B operand of the value product: the lane's registers 8 c .. 8 c + 7 are keys 16 c + 8 (i >> 2) + 4 hi + (i & 3),
    // the order the packed V rows are stored in
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float x8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x8[i] = pf[8 * c + i];
      bf16x8 pp[3];
      split3(x8, pp);
      mfma6(va[c], pp, o);
    }
    load_v(va, kt + 32);
  };
  int kt = t0;
  for (; kt + 64 < t1; kt += 32) {
    step(kt, sa, sb, std::false_type{});
    sa = sb;
  }
  if (kt + 32 < t1) {
    step(kt, sa, sb, std::false_type{});
    step(kt + 32, sb, sa, std::true_type{});
  } else if (kt < t1) {
    step(kt, sa, sb, std::true_type{});
  }

  // ---- merge of the four key quarters through LDS (fixed wave order: deterministic); identical to the fp32 kernel ----
  {
    const float lt = l + __shfl_xor(l, 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = o[r];
    red[wave][16][lane] = m;
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
  const long grow = qrow0 + qi;
  const int c = h * 32 + 8 * wave + 4 * hi;
  if (p.nsplit == 1) {
    const float inv = 1.f / lsum;
    acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    *reinterpret_cast<float4*>(p.out + grow * p.ldo + c) = acc;
  } else {
    const long rows = (long)p.B * p.Nq;
    *reinterpret_cast<float4*>(p.part + ((long)split * rows + grow) * C + c) = acc;
    if (wave == 0 && hi == 0) {
      float* ml = p.part + (long)p.nsplit * rows * C + (((long)split * rows + grow) * p.H + h) * 2;
      ml[0] = mm;
      ml[1] = lsum;
    }
  }
}

// ---- packing: x [B lanes][rows][ldx] fp32 -> the 16-byte operand chunks of a packed bank ---------------------------------------
// One thread per CHUNK (the eight values one lane contracts in one MFMA sub-step): it gathers its eight fp32 values, splits them
// and stores three 16-byte pieces (one per plane) -- coalesced on both sides.  K-style (TR = false): lane (row j of the tile, half
// hi) holds dims 16 c + 8 hi .. + 8 of bank row 32 tile + j.  V-style (TR = true): lane (channel j, half hi) holds the eight bank
// rows 32 tile + 16 c + 8 (i >> 2) + 4 hi + (i & 3), i = 0..7 -- the order the score tile's C/D layout keeps the keys in.  Rows
// are appended frame by frame at offsets that are no multiple of 8, so a V chunk can straddle two appends: the rows this call
// does not cover keep what the chunk holds (read-modify-write; zero in a fresh bank).
// Block (tile, nb) of a lane's bank sits at ((lane * cap_tiles + tile) * NB + nb) * blk_stride + blk_off ushorts (the d = 32 bank
// interleaves K and V blocks: stride 6144, V at offset 3072; the gated banks are separate: stride 3072).
template <bool TR>
__device__ __forceinline__ void pack_chunk(const float* __restrict__ x, unsigned short* __restrict__ planes, int B, long rows, int NB,
                                           long src_brows, int ldx, long cap_tiles, const int* __restrict__ slot_dev, int slot,
                                           int blk_stride, int blk_off, int ntiles_max) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long per = (long)ntiles_max * NB * 128;
  if (idx >= per * B) return;
  const int b = (int)(idx / per);
  long rem = idx - (long)b * per;
  const int lane = (int)(rem & 63), c = (int)((rem >> 6) & 1);
  rem >>= 7;
  const int nb = (int)(rem % NB);
  const long t0 = (long)(slot_dev ? *slot_dev : slot) * rows, t1 = t0 + rows;
  const long tile = (t0 >> 5) + rem / NB;
  if (tile > ((t1 - 1) >> 5)) return;
  const int j = lane & 31, hi = lane >> 5;
  unsigned short* dst = planes + (((long)b * cap_tiles + tile) * NB + nb) * blk_stride + blk_off + (c * 64 + lane) * 8;
  float xr[8];
  if (!TR) {
    const long row = tile * 32 + j;
    if (row < t0 || row >= t1) return;
    const float4* src = reinterpret_cast<const float4*>(x + ((long)b * src_brows + row - t0) * ldx + nb * 32 + 16 * c + 8 * hi);
    const float4 u0 = src[0], u1 = src[1];
    xr[0] = u0.x; xr[1] = u0.y; xr[2] = u0.z; xr[3] = u0.w; xr[4] = u1.x; xr[5] = u1.y; xr[6] = u1.z; xr[7] = u1.w;
  } else {
    const long r0 = tile * 32 + 16 * c + 4 * hi;
    unsigned covered = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long r = r0 + 8 * (i >> 2) + (i & 3);
      xr[i] = 0.f;
      if (r >= t0 && r < t1) {
        covered |= 1u << i;
        xr[i] = x[((long)b * src_brows + r - t0) * ldx + nb * 32 + j];
      }
    }
    if (!covered) return;
    if (covered != 0xffu) {        // rows of another append: rebuild their fp32 values from the planes the chunk holds
      u32x4 old[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) old[pl] = *reinterpret_cast<const u32x4*>(dst + pl * 1024);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (!(covered & (1u << i))) {
          float acc = 0.f;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            const unsigned w = old[pl][i >> 1];
            acc += __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));       // exact: the planes are the fp32 number's pieces
          }
          xr[i] = acc;
        }
    }
  }
  bf16x8 pl3[3];
  split3(xr, pl3);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * 1024) = __builtin_bit_cast(u32x4, pl3[pl]);
}

template <bool TR>
__global__ void __launch_bounds__(256) attn_pack_chunks_kernel(const float* __restrict__ x, unsigned short* __restrict__ planes, int B,
                                                               long rows, int NB, long src_brows, int ldx, long cap_tiles,
                                                               const int* __restrict__ slot_dev, int slot, int blk_stride,
                                                               int blk_off, int ntiles_max) {
  pack_chunk<TR>(x, planes, B, rows, NB, src_brows, ldx, cap_tiles, slot_dev, slot, blk_stride, blk_off, ntiles_max);
}
// K and V of the d = 32 bank in one launch: blockIdx.y = 0 packs k (K-style, block offset 0), 1 packs v (V-style, offset 3072)
__global__ void __launch_bounds__(256) attn_pack_kv_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                           unsigned short* __restrict__ planes, int B, long rows, int NB, long src_brows,
                                                           int ldk, int ldv, long cap_tiles, const int* __restrict__ slot_dev, int slot,
                                                           int ntiles_max) {
  if (blockIdx.y == 0) pack_chunk<false>(k, planes, B, rows, NB, src_brows, ldk, cap_tiles, slot_dev, slot, 6144, 0, ntiles_max);
  else pack_chunk<true>(v, planes, B, rows, NB, src_brows, ldv, cap_tiles, slot_dev, slot, 6144, 3072, ntiles_max);
}

static int launch_pack(bool tr, const float* x, unsigned short* planes, int B, long rows, int C, long src_brows, int ldx, long cap_rows,
                       const int* slot_dev, int slot, hipStream_t s) {
  const int NB = C >> 5;
  const int ntiles_max = (int)(rows / 32 + 2);
  const long n = (long)B * ntiles_max * NB * 128;
  if (tr)
    hipLaunchKernelGGL((attn_pack_chunks_kernel<true>), dim3(cdiv(n, 256)), dim3(256), 0, s, x, planes, B, rows, NB, src_brows, ldx,
                       cap_rows >> 5, slot_dev, slot, 3072, 0, ntiles_max);
  else
    hipLaunchKernelGGL((attn_pack_chunks_kernel<false>), dim3(cdiv(n, 256)), dim3(256), 0, s, x, planes, B, rows, NB, src_brows, ldx,
                       cap_rows >> 5, slot_dev, slot, 3072, 0, ntiles_max);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? AOT_OK : (int)e;
}

// ---- gated-propagation form (DeAOT, attention.py:672-707) in the bf16x6 family: the twin of attn_fwd_wide_coop_kernel<8> ----------
// One 128-wide query / key head, a value 1024 wide = [V | ID_V], the gate fused.  K planes [lane][tile][4 blocks][3072], V planes
// [lane][tile][32 blocks][3072] (aot_attn_pack_x6_part_f32).  Rounds 3-5 ran one 4-wave workgroup per 32 queries, every wave
// repeating the softmax: 0.23-0.32 of the bf16x6 roof.  Round 6 (below): 64 queries per workgroup, the softmax shared and
// software-pipelined -- 0.43 at a 14-frame bank, 0.39-0.41 over a clip's launch mix (profiles/r06_gated64_*.txt).
struct GatedX6Params {
  const float* q;
  const unsigned short* kp;
  const unsigned short* vp;
  const float* gate;
  float* out;
  float* part;
  const int* T_dev;
  int Nq, T, ldq, ldg, ldo, nsplit, B;
  long cap_rows;
  float scale_div;
};

// The score tile of the kernel must NOT live in accumulator registers: its 256 AGPRs are the 2 x 8 output blocks, and hipcc
// selects the AGPR form for every MFMA of a kernel that uses AGPRs at all (16 more would spill the output blocks around each score
// chain: 486 dwords of scratch in the first build).  The hardware takes a VGPR destination just as well, so the 24 score MFMAs of a
// key tile are written by hand with "v" constraints.  Hazards the compiler no longer sees (LLVM's gfx940 table, confirmed on the
// hardware in round 4, profiles/r04_mfma_hazard_probe.txt): dependent MFMAs on one accumulator issue back to back; a VALU / LDS
// store reading the result needs 12 wait states after the last MFMA of the chain -> mfma_vgpr_settle() (16 states, once per chain);
// the A / B operands come from loads or loop-invariant registers (s_waitcnt is placed from the register operands as for any asm).
__device__ __forceinline__ void mfma_bf16_v(const bf16x8& a, const bf16x8& b, f32x16& acc) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_bf16_v0(const bf16x8& a, const bf16x8& b, f32x16& acc) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
template <bool FIRST>
__device__ __forceinline__ void mfma6_vgpr(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& acc) {
  if (FIRST) mfma_bf16_v0(a[1], b[1], acc); else mfma_bf16_v(a[1], b[1], acc);
  mfma_bf16_v(a[0], b[2], acc);
  mfma_bf16_v(a[2], b[0], acc);
  mfma_bf16_v(a[0], b[1], acc);
  mfma_bf16_v(a[1], b[0], acc);
  mfma_bf16_v(a[0], b[0], acc);
}
__device__ __forceinline__ void mfma_vgpr_settle(f32x16& acc) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc)); }

// ---- the kernel: 64 queries per workgroup, software-pipelined ---------------------------------------------------------------------
// A workgroup of four waves (one per SIMD, 256 VGPRs + 256 AGPRs each) owns TWO 32-query tiles and a key range.  Wave w contracts
// channels [32 w, 32 w + 32) of q . k for both tiles (24 MFMAs per key tile; the partial score tiles meet in LDS in wave order) and
// owns value blocks [8 w, 8 w + 8) for both: every V fragment it fetches feeds the MFMAs of both tiles (the A operand stays in its
// registers, only the B operand P changes) -- 221 KB per key tile buy 4 x 216 MFMAs instead of 4 x 108.
// What the first builds of this shape taught (profiles/r06_gated64_first.txt, r06_gated64_pipelined.txt): with each wave doing its own
// two softmaxes + P splits before the value products a key tile took ~13 000 cycles for 6912 cycles of MFMA even with 27 workgroups
// alone on the chip -- one wave per SIMD issues in order, so ~700 vector instructions ran with the matrix pipe idle, and hipcc sank
// the V fetches to a few MFMAs before their use.  Hence:
//   * the softmax of key tile i + 1 runs INSIDE the value products of tile i.  The four waves share it: wave w sums the partial
//     score tiles of query tile w >> 1 and the running maximum (both waves of a pair compute the same bits), exponentiates and
//     splits only the eight score registers of sub-step w & 1, and publishes its three P planes (3 x 16 B per lane) and the rescale
//     factor through LDS; after the tile's barrier every wave reads the 12 plane fragments of both query tiles straight into MFMA
//     B operands.  ~130 vector instructions per wave and tile instead of ~700;
//   * no branch inside a tile: out-of-range rows are masked by select on every tile (a range's last tile and the look-ahead tiles
//     past it alike), so the whole tile is ONE scheduling region, cut by sched_barrier() into eight value blocks; each block's
//     V fetch (three blocks ahead, four rotating register sets, across the tile boundary) and its share of the softmax / of the
//     next-but-one tile's score MFMAs are pinned to it;
//   * Q planes parked in LDS, bank fetches through buffer descriptors (one per-lane offset register): no spill in the loop.
// Row sums: wave (t, c) keeps the sum of ITS eight registers; the four (sub-step, lane half) pieces of a query meet once, at the end.
// Block order: the grid is ONE dimension and XCD-major -- block id -> XCD id % 8 (the dispatcher's round robin), and the (lane,
// key range, query pair) triples are dealt so that each XCD gets a CONTIGUOUS run of them in key-range-major order: an XCD's L2
// streams one or two key ranges instead of all of them (R50-DeAOTL 422 against 412 fps with the ranges-fastest order).
// Counters at a 14-frame bank (profiles/r06_gated64_pmc.txt): matrix pipe 56 % busy on the SIMDs that have a wave, waves parked on
// memory 13 %; 25 B per clock and CU arrive from the L2 -- the rate every streaming kernel of this library sees.
constexpr int AOT_GX6_NVB = 4;      // rotating V register sets (2 and 4 measured equal once nothing spills)
template <int NVB>
__global__ void __launch_bounds__(256, 1) attn_x6_wide64p_kernel(const GatedX6Params p) {
  static_assert(8 % NVB == 0, "the rotating V sets must divide the eight value blocks of a tile");
  constexpr int NDV = 8, PS = 20;
  const int ntq = (p.Nq + 63) >> 6;
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
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 31, hi = lane >> 5;
  const int st = wave >> 1, sc_half = wave & 1;     // softmax role: query tile, sub-step
  const int T = p.T_dev ? *p.T_dev : p.T;
  const int ntile = (T + 31) >> 5;
  const int tps = (ntile + p.nsplit - 1) / p.nsplit;
  const int t0 = min(T, split * tps * 32);
  const int t1 = min(T, t0 + tps * 32);
  const long qrow0 = (long)b * p.Nq;
  const long cap_tiles = p.cap_rows >> 5;
  __shared__ float part[2][2][4][64 * PS];        // [buffer][query tile][wave][lane][score register (+ pad)]
  __shared__ u32x4 pbuf[2][2][2][3][64];           // [buffer][query tile][sub-step][plane][lane]: P^T as bf16 B-operand fragments
  __shared__ float abuf[2][2][64];                 // [buffer][query tile][lane]: factor the accumulators take before that tile
  __shared__ float lbuf[2][2][64], mbuf[2][64];    // end of the range: row-sum pieces [query tile][sub-step][lane], maxima

  // Q^T planes of this wave's 32 channels, both query tiles: 12 fragments of 16 B per lane, parked in LDS (they are needed for 24 of
  // a tile's 216 MFMAs; as registers they would cost the fourth V set)
  __shared__ u32x4 qbuf[4][2][2][3][64];           // [wave][query tile][sub-step][plane][lane]
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qrow = min(qt * 64 + t * 32 + j, p.Nq - 1);
    const float* src = p.q + (qrow0 + qrow) * p.ldq + wave * 32 + hi * 8;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float4 u0 = *reinterpret_cast<const float4*>(src + 16 * c), u1 = *reinterpret_cast<const float4*>(src + 16 * c + 4);
      float x[8] = {u0.x / p.scale_div, u0.y / p.scale_div, u0.z / p.scale_div, u0.w / p.scale_div,
                    u1.x / p.scale_div, u1.y / p.scale_div, u1.z / p.scale_div, u1.w / p.scale_div};
      bf16x8 q3[3];
      split3(x, q3);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) qbuf[wave][t][c][pl][lane] = __builtin_bit_cast(u32x4, q3[pl]);
    }
  }
  // bank fetches through buffer descriptors: ONE per-lane offset register (lane * 16), the (tile, block) part a scalar, the 1 KB
  // chunk an immediate -- as 64-bit global addresses hipcc kept a dozen VGPR pairs alive for them.  The descriptors are based at
  // the range's first tile of this wave's blocks (64-bit pointer arithmetic), so their 32-bit offsets see the range only
  // (< 4 GB: checked by the entry point); tiles are clamped to the bank's capacity, never to the descriptor's size.
  const long tile0 = t0 >> 5, tile_last = cap_tiles - 1 - tile0;      // (relative) last tile that exists
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(p.kp) + (((long)b * cap_tiles + tile0) * 4 + wave) * 3072, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(p.vp) + (((long)b * cap_tiles + tile0) * 32 + wave * NDV) * 3072, 0, 0xffffffff, 0x00020000);
  const int lane_off = lane * 16;

  float m = -INFINITY, l = 0.f;       // of query tile st, over the registers of sub-step sc_half
  f32x16 o[2][NDV];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int d = 0; d < NDV; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][d][r] = 0.f;

  auto load_k = [&](bf16x8 (&kf)[2][3], int kt) {      // kt: key offset from t0
    const int so = (int)min((long)(kt >> 5), tile_last) * (4 * 6144);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        kf[c][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rk, lane_off + (pl * 2 + c) * 1024, so, 0));
  };
  auto load_v = [&](bf16x8 (&vf)[2][3], int kt, int d) {
    const int so = ((int)min((long)(kt >> 5), tile_last) * 32 + d) * 6144;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        vf[c][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, lane_off + (pl * 2 + c) * 1024, so, 0));
  };
  // partial score tile (this wave's 32 channels) of query tile t for the keys in kf -> LDS
  auto qk_part = [&](const bf16x8 (&kf)[2][3], int buf, int t) {
    bf16x8 qf[2][3];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) qf[c][pl] = __builtin_bit_cast(bf16x8, qbuf[wave][t][c][pl][lane]);
    f32x16 sc;
    mfma6_vgpr<true>(kf[0], qf[0], sc);
    mfma6_vgpr<false>(kf[1], qf[1], sc);
    mfma_vgpr_settle(sc);
    float4* dst = reinterpret_cast<float4*>(&part[buf][t][wave][lane * PS]);
#pragma unroll
    for (int g = 0; g < 4; ++g) dst[g] = make_float4(sc[4 * g], sc[4 * g + 1], sc[4 * g + 2], sc[4 * g + 3]);
  };
  // this wave's share of the softmax of the tile starting at key kt (scores in part[buf]) -> pbuf[buf], abuf[buf]; in three pieces
  // so that the caller can spread them over value blocks
  float sm_sc[16], sm_p[8];
  auto softmax_a = [&](int buf, int kt) {      // sum of the four partial tiles, mask, maximum
#pragma unroll
    for (int g = 0; g < 4; ++g) {      // fixed wave order: both waves of the pair get the same bits
      const float4 a0 = *reinterpret_cast<const float4*>(&part[buf][st][0][lane * PS + 4 * g]);
      const float4 a1 = *reinterpret_cast<const float4*>(&part[buf][st][1][lane * PS + 4 * g]);
      const float4 a2 = *reinterpret_cast<const float4*>(&part[buf][st][2][lane * PS + 4 * g]);
      const float4 a3 = *reinterpret_cast<const float4*>(&part[buf][st][3][lane * PS + 4 * g]);
      sm_sc[4 * g] = ((a0.x + a1.x) + a2.x) + a3.x;
      sm_sc[4 * g + 1] = ((a0.y + a1.y) + a2.y) + a3.y;
      sm_sc[4 * g + 2] = ((a0.z + a1.z) + a2.z) + a3.z;
      sm_sc[4 * g + 3] = ((a0.w + a1.w) + a2.w) + a3.w;
    }
    const int rem = t1 - kt - 4 * hi;      // rows r of this lane with mfma32_row(r, 0) >= rem are past the range (rem <= 0: all)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (mfma32_row(r, 0) >= rem) sm_sc[r] = -INFINITY;
  };
  auto softmax_b = [&](int buf) {      // new maximum, rescale factor, the eight weights of this wave's sub-step, their sum
    const float x = max3f(max3f(max3f(sm_sc[0], sm_sc[1], sm_sc[2]), max3f(sm_sc[3], sm_sc[4], sm_sc[5]), max3f(sm_sc[6], sm_sc[7], sm_sc[8])),
                          max3f(sm_sc[9], sm_sc[10], sm_sc[11]), max3f(sm_sc[12], sm_sc[13], max3f(sm_sc[14], sm_sc[15], sm_sc[15])));
    const float mnew = fmaxf(m, fmaxf(x, __shfl_xor(x, 32)) * AOT_LOG2E);
    const float alpha = (m == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m - mnew);      // (m = -inf: nothing accumulated yet)
    m = mnew;
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float sv = sc_half ? sm_sc[8 + i] : sm_sc[i];
      sm_p[i] = (mnew == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(fmaf(sv, AOT_LOG2E, -mnew));
      ps += sm_p[i];
    }
    l = l * alpha + ps;
    abuf[buf][st][lane] = alpha;      // (both waves of the pair store the same value)
  };
  auto softmax_c = [&](int buf) {      // the three bf16 planes of the eight weights -> LDS
    bf16x8 pl3[3];
    split3(sm_p, pl3);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) pbuf[buf][st][sc_half][pl][lane] = __builtin_bit_cast(u32x4, pl3[pl]);
  };

  bf16x8 ka[2][3], vb[NVB][2][3];
  const int nt = (t1 - t0 + 31) >> 5;       // key tiles of this range
  if (nt <= 0) {                            // empty range (more splits than tiles): an all-zero partial with m = -inf
    // falls through to the epilogue with o = 0, l = 0, m = -inf
  }
  // prologue: scores of tile 0 -> softmax of tile 0 -> scores of tile 1; V of the first NVB - 1 blocks on their way
  load_k(ka, 0);
#pragma unroll
  for (int d = 0; d < NVB - 1; ++d) load_v(vb[d], 0, d);
  qk_part(ka, 0, 0);
  qk_part(ka, 0, 1);
  load_k(ka, 32);
  __syncthreads();
  softmax_a(0, t0);
  softmax_b(0);
  softmax_c(0);
  qk_part(ka, 1, 0);
  qk_part(ka, 1, 1);
  __syncthreads();

  for (int i = 0; i < nt; ++i) {
    const int kt = t0 + 32 * i, buf = i & 1;
    // P of this tile (both query tiles) and the factors its accumulators take first
    bf16x8 pp[2][2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) pp[t][c][pl] = __builtin_bit_cast(bf16x8, pbuf[buf][t][c][pl][lane]);
    const float al0 = abuf[buf][0][lane], al1 = abuf[buf][1][lane];
    if (__any(al0 != 1.f || al1 != 1.f)) {      // (rare after the first tiles of a range)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int d = 0; d < NDV; ++d)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            float a0, a1;
            asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(a0), "=v"(a1) : "a"(o[t][d][r]), "a"(o[t][d][r + 1]));
            a0 *= t ? al1 : al0;
            a1 *= t ? al1 : al0;
            float w0, w1;
            asm volatile("v_accvgpr_write_b32 %0, %2\n\tv_accvgpr_write_b32 %1, %3" : "=a"(w0), "=a"(w1) : "v"(a0), "v"(a1));
            o[t][d][r] = w0;
            o[t][d][r + 1] = w1;
          }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < NDV; ++d) {
      // the block NVB - 1 ahead: of this tile, or of the next one (clamped past the bank's end: fetched, never used)
      {
        const int dn = d + NVB - 1;
        if (dn < NDV) load_v(vb[dn % NVB], 32 * i, dn);
        else load_v(vb[dn % NVB], 32 * i + 32, dn - NDV);
      }
      // this block's share of the look-ahead work
      if (d == 1) softmax_a(buf ^ 1, kt + 32);
      if (d == 2) softmax_b(buf ^ 1);
      if (d == 3) { softmax_c(buf ^ 1); load_k(ka, 32 * i + 64); }
      if (d == 5) qk_part(ka, buf, 0);            // scores of tile i + 2 into the buffer tile i's softmax has finished with
      if (d == 6) qk_part(ka, buf, 1);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        mfma6(vb[d % NVB][c], pp[0][c], o[0][d]);
        mfma6(vb[d % NVB][c], pp[1][c], o[1][d]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // ---- end of the range: the four row-sum pieces and the maximum of each query tile meet ----
  lbuf[st][sc_half][lane] = l;
  mbuf[st][lane] = m;
  __syncthreads();
  const long prow = (long)p.B * p.Nq;
  const int cbase = wave * 32 * NDV + 4 * hi;
  constexpr int CV = 32 * NDV * 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float lh = lbuf[t][0][lane] + lbuf[t][1][lane];
    const float lt = lh + __shfl_xor(lh, 32);
    const float mt = mbuf[t][lane];
    const int ql = qt * 64 + t * 32 + j;
    if (ql >= p.Nq) continue;
    const long qi = qrow0 + ql;
    if (p.nsplit == 1) {
      const float inv = 1.f / lt;
#pragma unroll
      for (int d = 0; d < NDV; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v4 = make_float4(o[t][d][4 * g] * inv, o[t][d][4 * g + 1] * inv, o[t][d][4 * g + 2] * inv, o[t][d][4 * g + 3] * inv);
          const int c = cbase + d * 32 + 8 * g;
          if (p.gate) {
            const float4 u = *reinterpret_cast<const float4*>(p.gate + qi * p.ldg + c);
            v4.x *= u.x; v4.y *= u.y; v4.z *= u.z; v4.w *= u.w;
          }
          *reinterpret_cast<float4*>(p.out + qi * p.ldo + c) = v4;
        }
    } else {
      float* dst = p.part + ((long)split * prow + qi) * CV;
#pragma unroll
      for (int d = 0; d < NDV; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(dst + cbase + d * 32 + 8 * g) =
              make_float4(o[t][d][4 * g], o[t][d][4 * g + 1], o[t][d][4 * g + 2], o[t][d][4 * g + 3]);
      if (hi == 0) {
        float* ml = p.part + (long)p.nsplit * prow * CV + (((long)split * prow + qi) * 4 + wave) * 2;
        ml[0] = mt;
        ml[1] = lt;
      }
    }
  }
}

}  // namespace

// ===== C ABI ==============================================================================================================
extern "C" int aot_attn_pack_x6_f32(const float* k, const float* v, void* kv, int B, long rows, int C, long src_brows, int ldk,
                                    int ldv, long cap_rows, const int* slot_dev, int slot, void* stream) {
  if (!k || !v || !kv || B <= 0 || rows <= 0 || C <= 0 || (C & 31) || (ldk & 3) || (ldv & 3) || cap_rows <= 0 ||
      (cap_rows & 31) || slot < 0 || src_brows < 0 || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)kv & 15))
    return AOT_ERR_BADARG;
  if (!slot_dev && ((long)slot + 1) * rows > cap_rows) return AOT_ERR_BADARG;
  const int NB = C >> 5, ntiles_max = (int)(rows / 32 + 2);
  const long n = (long)B * ntiles_max * NB * 128;
  hipLaunchKernelGGL(attn_pack_kv_kernel, dim3(cdiv(n, 256), 2), dim3(256), 0, (hipStream_t)stream, k, v, (unsigned short*)kv, B, rows, NB,
                     src_brows, ldk, ldv, cap_rows >> 5, slot_dev, slot, ntiles_max);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_attn_x6_f32(const float* q, const void* kv, float* out, float* part, int B, long cap_rows, int Nq, int T,
                               const int* T_dev, int H, int d, int ldq, int ldo, float scale_div, int nsplit, void* stream) {
  if (d != 32) return AOT_ERR_UNSUPPORTED;
  if (!q || !kv || !out || Nq <= 0 || T <= 0 || H <= 0 || B <= 0 || cap_rows < T || (cap_rows & 31)) return AOT_ERR_BADARG;
  if ((long)B * cdiv(Nq, 32) > 65535) return AOT_ERR_UNSUPPORTED;
  if ((ldq & 3) || (ldo & 3) || ((uintptr_t)q & 15) || ((uintptr_t)out & 15) || ((uintptr_t)kv & 15))
    return AOT_ERR_BADARG;
  if (nsplit < 1 || (nsplit > 1 && !part)) return AOT_ERR_BADARG;
  AttnX6Params p;
  p.q = q; p.kv = (const unsigned short*)kv; p.out = out; p.part = part; p.T_dev = T_dev;
  p.Nq = Nq; p.T = T; p.H = H; p.ldq = ldq; p.ldo = ldo; p.nsplit = nsplit; p.B = B; p.cap_rows = cap_rows;
  p.scale_div = scale_div;
  hipLaunchKernelGGL(attn_x6_d32_kernel, dim3(H, B * cdiv(Nq, 32), nsplit), dim3(256), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_attn_pack_x6_part_f32(const float* x, void* planes, int B, long rows, int C, long src_brows, int ldx, long cap_rows,
                                         const int* slot_dev, int slot, int transpose, void* stream) {
  if (!x || !planes || B <= 0 || rows <= 0 || C <= 0 || (C & 31) || (ldx & 3) || cap_rows <= 0 || (cap_rows & 31) || slot < 0 ||
      src_brows < 0 || ((uintptr_t)x & 15) || ((uintptr_t)planes & 15))
    return AOT_ERR_BADARG;
  if (!slot_dev && ((long)slot + 1) * rows > cap_rows) return AOT_ERR_BADARG;
  return launch_pack(transpose != 0, x, (unsigned short*)planes, B, rows, C, src_brows, ldx, cap_rows, slot_dev, slot, (hipStream_t)stream);
}

extern "C" int aot_gated_attn_x6_f32(const float* q, const void* kp, const void* vp, const float* gate, float* out, float* part, int B,
                                     long cap_rows, int Nq, int T, const int* T_dev, int dqk, int dv, int ldq, int ldg, int ldo,
                                     float scale_div, int nsplit, void* stream) {
  if (dqk != 128 || dv != 1024) return AOT_ERR_UNSUPPORTED;
  if (!q || !kp || !vp || !out || Nq <= 0 || T <= 0 || B <= 0 || cap_rows < T || (cap_rows & 31)) return AOT_ERR_BADARG;
  if ((ldq & 3) || (ldo & 3) || (gate && (ldg & 3)) || ((uintptr_t)q & 15) || ((uintptr_t)out & 15) || ((uintptr_t)kp & 15) ||
      ((uintptr_t)vp & 15))
    return AOT_ERR_BADARG;
  if (nsplit < 1 || (nsplit > 1 && !part)) return AOT_ERR_BADARG;
  // the kernel addresses a key range through 32-bit buffer offsets (196 608 B of V planes per key tile)
  if (((cap_rows >> 5) / nsplit + 2) * 196608L > 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  GatedX6Params p;
  p.q = q; p.kp = (const unsigned short*)kp; p.vp = (const unsigned short*)vp; p.out = out; p.part = part; p.T_dev = T_dev;
  p.gate = (nsplit == 1) ? gate : nullptr;     // with splits the gate is applied by aot_attn_merge_f32
  p.Nq = Nq; p.T = T; p.ldq = ldq; p.ldg = ldg; p.ldo = ldo; p.nsplit = nsplit; p.B = B; p.cap_rows = cap_rows; p.scale_div = scale_div;
  const int total = B * nsplit * cdiv(Nq, 64);
  hipLaunchKernelGGL((attn_x6_wide64p_kernel<AOT_GX6_NVB>), dim3(8 * cdiv(total, 8)), dim3(256), 0, (hipStream_t)stream, p);
  AOT_LAUNCH_CHECK();
}
