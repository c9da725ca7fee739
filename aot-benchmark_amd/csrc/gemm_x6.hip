// The bf16x6 conv / linear family (mfma = 'bf16x6'): fp32-equivalent arithmetic on the bf16 matrix cores of gfx950, plus the plain bf16
// member of the training path.  (Until round 6 these kernels lived in gemm_lds.hip next to the fp32 family.)
#include "gemm_tile.h"

namespace {

// ---- bf16 x 6 variant (second, parity-gated kernel family) --------------------------------------------------------------
// fp32 MFMA runs at 1/16 of the bf16 rate on gfx950.  Every fp32 number is EXACTLY the sum of three bf16 numbers obtained by
// truncation (8 + 8 + 8 significand bits: hi = x & 0xffff0000, mid = (x - hi) & 0xffff0000, lo = x - hi - mid), so
//   a * w = sum of the nine products (a_i * w_j);   the six of order i + j <= 2 are kept (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi,
//   mid*mid): what is dropped is <= 3 * 2^-24 |a w| -- the size of one fp32 rounding of the product.  Each kept product of two
//   bf16 values is exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, the sums are fp32: an fp32-equivalent GEMM at
//   6/16 of the fp32-MFMA cost (measured in registers: 268 vs 150 TFLOP/s-equivalent, profiles/r02_bf16_split_rate.txt;
//   parity by emulation through the oracle: 1.8e-5 on the logits of BASELINE config 2, the fp32 path's own level).
// Same 64x64 tile, item walk, LDS-DMA ring and tile end as gemm_lean_kernel<1>.  What differs:
//   * the WEIGHT comes pre-split (aot_pack_bf16x6, once per model): three bf16 planes in a tile-friendly order,
//     w6[plane][K/32][4][Cout_pad][8]: the 16-byte chunk cc = 2*s + h of a 32-wide k-block holds the eight k values lane-half h
//     contracts in sub-step s (k = 16 s + 4 h + {0..3} and + 8), for Cout_pad (a multiple of 64) columns side by side -- one
//     LDS-DMA piece is 64 columns x 16 bytes, contiguous in memory AND lane-linear in LDS, so the fragment reads
//     (ds_read_b128, consecutive lanes = consecutive columns) are conflict-free without a swizzle;
//   * the ACTIVATION tile arrives as fp32 exactly as in the lean kernel (im2col through the buffer descriptor) and is split in
//     registers right before use: 4 VALU per element + 3 v_perm per pair;
//   * three ring stages of 20.6 KB (two workgroups per CU); a k-step is 12 MFMAs of 32 cycles instead of 16 of 64.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct X6Weight {
  const void* w6;     // [3][K/32][4][cout_pad][8] bf16
  int cout_pad;       // multiple of 64
};

// two truncated bf16 (the upper halves of a and b) in one dword: [a.hi16 | b.hi16 << 16]
__device__ __forceinline__ unsigned pack_hi16(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// eight fp32 values (the lane's k-set of one sub-step) -> their three bf16 planes
__device__ __forceinline__ void split3(const f32x4& x0, const f32x4& x1, bf16x8 (&out)[3]) {
  u32x4 w[3];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float r0 = e < 2 ? x0[2 * e] : x1[2 * e - 4], r1 = e < 2 ? x0[2 * e + 1] : x1[2 * e - 3];
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

// asynchronous fragment reads of one ring stage (IMM = its byte offset): four fp32 chunks of the lane's A row, and for each weight
// plane the lane's two bf16 chunk columns (sub-steps 0 / 1 = pieces 4 pl + half and 4 pl + 2 + half; `half` is in baddr)
template <int IMM, int PIECE, int NT = 6>
__device__ __forceinline__ void x6_fetch(f32x4 (&a)[4], bf16x8 (&b)[3][2], const unsigned (&aaddr)[4], unsigned baddr) {
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[j]) : "v"(aaddr[j]), "n"(IMM));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[0][0]) : "v"(baddr), "n"(IMM + 0 * PIECE));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[0][1]) : "v"(baddr), "n"(IMM + 2 * PIECE));
  if (NT == 1) return;        // plain bf16: the first plane is the whole (rounded) weight
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[1][0]) : "v"(baddr), "n"(IMM + 4 * PIECE));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[1][1]) : "v"(baddr), "n"(IMM + 6 * PIECE));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[2][0]) : "v"(baddr), "n"(IMM + 8 * PIECE));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[2][1]) : "v"(baddr), "n"(IMM + 10 * PIECE));
}
template <int NT = 6>
__device__ __forceinline__ void x6_landed(f32x4 (&a)[4], bf16x8 (&b)[3][2]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#pragma unroll
  for (int pl = 0; pl < (NT == 1 ? 1 : 3); ++pl) asm volatile("" : "+v"(b[pl][0]), "+v"(b[pl][1]));
}

// eight fp32 values -> eight bf16, round to nearest even (v_cvt_pk_bf16_f32, gfx950), element order as split3
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 round8(const f32x4& x0, const f32x4& x1) {
  u32x4 w;
  w[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{x0[0], x0[1]}, hbf16x2));
  w[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{x0[2], x0[3]}, hbf16x2));
  w[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{x1[0], x1[1]}, hbf16x2));
  w[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{x1[2], x1[3]}, hbf16x2));
  return __builtin_bit_cast(bf16x8, w);
}

// The plain bf16 member (the training path's `precision = 'bf16'`, train_ops.matmul_precision): ONE product of operands rounded to
// bf16 -- weight plane from aot_pack_bf16_f32 (round to nearest even), activations rounded in registers; 2 MFMAs per k-step, one
// weight plane through the LDS-DMA ring.  (Rounds 3-4 ran the six-term inference product on this tile walk too -- NT = 6, and a form on
// pre-split activation planes; the register-staged kernels below replaced them in round 5 and round 6 removed them.)
// SK (1x1 only): split-K for the weight gradients -- item = (tile, k-slice), the partial tile goes raw to its fp32 slab of `scratch`
// [ksplit][M][Cout] and splitk_reduce_kernel sums the slabs in order (+ bias / residual / act).
template <bool IS1X1, bool SK = false>
__global__ void __launch_bounds__(256, 2) gemm_bf16_kernel(const ConvParams p, const X6Weight wq, const int ksplit, float* __restrict__ scratch) {
  static_assert(!SK || IS1X1, "split-K: the 1x1 member only");
  constexpr int NT = 1;
  constexpr int NSTORE = 16;                            // stores of a tile end per lane
  constexpr int NST = 3;
  constexpr int BM = 64, BN = 64;
  constexpr int AG = BM / 8, AGW = AG / 4;              // A: 8-row fp32 groups, two per wave
  constexpr int BPW = 1;                                // B: one 16-byte chunk column (cc = wave) of the plane per wave
  constexpr int LPW = AGW + BPW;
  constexpr int AEL = 4;                                // bytes per activation element
  constexpr int OPA_BYTES = AG * GROUP_STRIDE, B_PIECE = 64 * 16, OPB_BYTES = 12 * B_PIECE;
  constexpr int STAGE_BYTES = OPA_BYTES + OPB_BYTES;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int nbn = (p.Cout + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
  const int nk = SK ? (p.K / BK) / ksplit : p.K / BK;            // k-steps per item (host guarantees divisibility)
  const int nitems = nbm * nbn * (SK ? ksplit : 1);
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;          // XCD-aware item order, as in gemm_lds_kernel
  const int nwg_x = ((int)gridDim.x - xcd + 7) >> 3;
  const int q8 = nitems >> 3, r8 = nitems & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
  const int mine = li < chunk_n ? (chunk_n - li + nwg_x - 1) / nwg_x : 0;
  const int total = mine * nk;
  if (total == 0) return;
  auto item_of = [&](int i) __attribute__((always_inline)) {
    const int it = chunk0 + li + i * nwg_x;
    Item r;
    r.bn = it % nbn;
    if (SK) {            // item = ((bm * ksplit) + slice) * nbn + bn, as in gemm_lean_kernel
      const int t = it / nbn;
      r.kt0 = (t % ksplit) * nk;
      r.bm = t / ksplit;
    } else {
      r.bm = it / nbn;
      r.kt0 = 0;
    }
    return r;
  };
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int lr = lane >> 3, lp = lane & 7;
  const int cofs = (lp ^ lr) << 2;
  const int hw_out = p.OH * p.OW;
  const int plane_bytes = (p.K / 8) * wq.cout_pad * 16;
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)((long)p.B * p.H * p.W * p.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wq.w6), 0, BPW * plane_bytes, 0x00020000);
  const i32x4 desc_out = raw_desc(p.out, (long)p.M * p.ldc * 4);
  const i32x4 desc_res = raw_desc(p.res, (long)(p.res_rows ? p.res_rows : p.M) * p.ldr * 4);
  const i32x4 desc_bias = raw_desc(p.bias, (long)p.Cout * 4);
  const int n_res = (!SK && p.res) ? 16 : 0, n_bias = (!SK && p.bias) ? 1 : 0;      // (split-K: the reduce pass adds them)

  // ---- issue side --------------------------------------------------------------------------------------------------
  int is_i = 0, is_kt = 0;
  int a_off[AGW], a_iy0[AGW], a_ix0[AGW];
  bool a_ok[AGW];
  unsigned b_off = 0;
  int s_k = 0, s_kb = 0;       // wave-uniform byte offsets along K: A rows of a 1x1 layer / the weight's k-blocks
  int tap_c = 0, tap_ky = 0, tap_kx = 0, s_tap = 0;
  auto setup_item = [&](int i) __attribute__((always_inline)) {
    const bool live = i < mine;
    const Item it = item_of(live ? i : 0);
#pragma unroll
    for (int g = 0; g < AGW; ++g) {
      // lane = (row lr of the wave's g-th 8-row group, 16-byte chunk lp)
      const int rowl = 8 * (AGW * wave + g) + lr;
      const int m = it.bm * BM + rowl;
      a_ok[g] = live && m < p.M;
      const int mm = a_ok[g] ? m : 0;
      const int b = mm / hw_out, pix = mm - b * hw_out;
      const int oy = pix / p.OW, ox = pix - oy * p.OW;
      a_iy0[g] = oy * p.stride - p.pad;
      a_ix0[g] = ox * p.stride - p.pad;
      a_off[g] = (((b * p.H + a_iy0[g]) * p.W + a_ix0[g]) * p.lda + cofs) * AEL;
      if (IS1X1 && !a_ok[g]) a_off[g] = (int)OOB;
    }
    b_off = live ? (unsigned)((wave * wq.cout_pad + it.bn * BN + lane) * 16) : OOB;    // chunk column cc = wave of k-block 0
    s_k = SK ? it.kt0 * BK * AEL : 0;                    // (split-K: the slice's first k-step)
    s_kb = SK ? it.kt0 * 4 * wq.cout_pad * 16 : 0;
    if (!IS1X1) { tap_c = 0; tap_ky = 0; tap_kx = 0; }
  };
  auto issue = [&](auto SLOT) __attribute__((always_inline)) -> void {
    constexpr int slot = decltype(SLOT)::value;
    if (is_kt == 0) setup_item(is_i);
    if (!IS1X1) s_tap = ((tap_ky * p.dil * p.W + tap_kx * p.dil) * p.lda + tap_c) * AEL;
    unsigned char* st = lds + slot * STAGE_BYTES;
    {
#pragma unroll
      for (int g = 0; g < AGW; ++g) {
        unsigned char* dst = st + (AGW * wave + g) * GROUP_STRIDE;
        if (IS1X1) {
          dma16(rsrc_a, dst, a_off[g], s_k);
        } else {
          const int iy = a_iy0[g] + tap_ky * p.dil, ix = a_ix0[g] + tap_kx * p.dil;
          const bool in = a_ok[g] & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
          dma16(rsrc_a, dst, in ? a_off[g] + s_tap : (int)OOB, 0);
        }
      }
    }
#pragma unroll
    for (int pl = 0; pl < BPW; ++pl)
      dma16(rsrc_b, st + OPA_BYTES + (pl * 4 + wave) * B_PIECE, (int)b_off, s_kb + pl * plane_bytes);
    s_k += BK * AEL;
    s_kb += 4 * wq.cout_pad * 16;
    if (!IS1X1) {
      tap_c += BK;
      if (tap_c >= p.Cin) { tap_c = 0; if (++tap_kx == p.KW) { tap_kx = 0; ++tap_ky; } }
    }
    if (++is_kt == nk) { is_kt = 0; ++is_i; }
  };

  // ---- compute side ------------------------------------------------------------------------------------------------
  const unsigned lds_base = (unsigned)(size_t)(lptr_t)lds;
  unsigned aaddr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) aaddr[j] = lds_base + chunk_off(wm + l31, 2 * j + half);
  const unsigned baddr = lds_base + OPA_BYTES + (half * 64 + wn + l31) * 16;      // chunk column cc = 2 s + half: + 2 s pieces
  f32x4 ra[2][4];              // [register set][16-byte chunk j]: sub-step s contracts chunks 2 s and 2 s + 1
  bf16x8 rb[2][3][2];          // [register set][plane][sub-step]
  auto fetch = [&](auto SET, auto SLOT) __attribute__((always_inline)) -> void {
    x6_fetch<decltype(SLOT)::value * STAGE_BYTES, B_PIECE, NT>(ra[decltype(SET)::value], rb[decltype(SET)::value], aaddr, baddr);
  };
  auto landed = [&](auto SET) __attribute__((always_inline)) -> void {
    x6_landed<NT>(ra[decltype(SET)::value], rb[decltype(SET)::value]);
  };
  f32x16 acc[2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  int c_i = 0, c_kt = 0;
  int stores_pending = 0;
  float rv[16], bv = 0.f;
  auto epi_loads = [&]() __attribute__((always_inline)) {
    const Item it = item_of(c_i);
    const int n = it.bn * BN + wn + l31;
    const bool col_ok = n < p.Cout;
    const int m0 = it.bm * BM;
    if (n_bias) bv = buf_load(desc_bias, col_ok ? n * 4 : (int)OOB);
    if (n_res && p.res_rows == 0) {
      const int mlane = m0 + wm + 4 * half;
      const int vbase = col_ok ? (mlane * p.ldr + n) * 4 : (int)OOB;
      const int rows_left = p.M - mlane, ldr4 = p.ldr * 4;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2);
        rv[r] = buf_load_s(desc_res, c < rows_left ? vbase : (int)OOB, c * ldr4);
      }
    } else if (n_res) {
      const int rr0 = m0 % p.res_rows;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dm = wm + mfma32_row(r, half);
        int rr = rr0 + dm;
        if (p.res_rows >= BM) rr = rr >= p.res_rows ? rr - p.res_rows : rr;
        else rr %= p.res_rows;
        rv[r] = buf_load(desc_res, (col_ok && m0 + dm < p.M) ? (rr * p.ldr + n) * 4 : (int)OOB);
      }
    }
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    const Item it = item_of(c_i);
    const int n = it.bn * BN + wn + l31;
    const bool col_ok = n < p.Cout;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] += acc[1][r]; acc[1][r] = 0.f; }
    if (SK) {            // the raw partial tile -> the slice's slab [M][Cout] (sixteen stores, counted like the fused form's)
      const int mlane_s = it.bm * BM + wm + 4 * half;
      const i32x4 desc_slab = raw_desc(scratch + (long)(it.kt0 / nk) * p.M * p.Cout, (long)p.M * p.Cout * 4);
      const int vbase_s = col_ok ? (mlane_s * p.Cout + n) * 4 : (int)OOB;
      const int rows_left_s = p.M - mlane_s, lds4 = p.Cout * 4;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2);
        buf_store_s(desc_slab, c < rows_left_s ? vbase_s : (int)OOB, c * lds4, acc[0][r]);
        acc[0][r] = 0.f;
      }
      stores_pending = 16;
      return;
    }
    // residual and bias were fetched under the tile's last k-step: older than the DMA pieces issued in that step
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW, 15));
    if (n_bias) asm volatile("" : "+v"(bv));
    if (n_res) {
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(rv[r]));
    }
    const int m0 = it.bm * BM;
    const int mlane = m0 + wm + 4 * half;
    const int vbase = col_ok ? (mlane * p.ldc + n) * 4 : (int)OOB;
    const int rows_left = p.M - mlane, ldc4 = p.ldc * 4;
    if (n_bias) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] += bv;
    }
    if (n_res) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] += rv[r];
    }
    with_act(p.act, [&](auto ACT) __attribute__((always_inline)) -> void {
      constexpr int act = decltype(ACT)::value;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2);
        buf_store_s(desc_out, c < rows_left ? vbase : (int)OOB, c * ldc4, apply_act(acc[0][r], act));
        acc[0][r] = 0.f;
      }
    });
    stores_pending = 16;
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  issue(I0{});
  issue(I1{});
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW, 15));          // step 0 has landed
  __builtin_amdgcn_s_barrier();
  fetch(I0{}, I0{});
  // step ss (ring stage U % 3, register set U % 2; the loop is unrolled by six): on entry the fragments of step ss are being
  // read into set U % 2, the DMA of step ss+1 is in flight
  auto step = [&](auto U) __attribute__((always_inline)) -> void {
    constexpr int u = decltype(U)::value, set = u & 1, nslot = (u + 1) % 3, islot = (u + 2) % 3;
    // step ss+1 has landed, and this wave's fragment reads of step ss (plus the stores of a tile the previous step finished)
    if (stores_pending) {
      __builtin_amdgcn_s_waitcnt(waitcnt_imm(NSTORE, 0));
      stores_pending = 0;
    } else {
      __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
    }
    __builtin_amdgcn_s_barrier();                          // ... for every wave; the stage of step ss-1 (= of step ss+2) is free
    landed(std::integral_constant<int, set>{});
    if (c_kt == nk - 1) epi_loads();     // last k-step of the tile: its residual and bias, now -- BEFORE this step's DMA pieces, so
                                         // that the epilogue's counted wait (all but the youngest LPW) covers them
    fetch(std::integral_constant<int, set ^ 1>{}, std::integral_constant<int, nslot>{});      // fragments of step ss+1
    issue(std::integral_constant<int, islot>{});                                             // DMA of step ss+2
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(round8(ra[set][2 * s], ra[set][2 * s + 1]), rb[set][0][s], acc[s], 0, 0, 0);
    }
    if (++c_kt == nk) {
      epilogue();
      c_kt = 0;
      ++c_i;
    }
  };
#pragma unroll 1
  for (int ss = 0; ss < total; ss += 6) {
    step(I0{});
    if (ss + 1 < total) step(I1{});
    if (ss + 2 < total) step(I2{});
    if (ss + 3 < total) step(std::integral_constant<int, 3>{});
    if (ss + 4 < total) step(std::integral_constant<int, 4>{});
    if (ss + 5 < total) step(std::integral_constant<int, 5>{});
  }
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));           // the all-out-of-bounds DMAs past the end still target this LDS
}

// ---- register-staged 64x64 tile ("x6r", round 5) -------------------------------------------------------------------------------
// What the round-5 probes of the LDS-DMA kernels say (profiles/r05_x6pp_probes.txt): a k-step of those kernels is set by the ISSUE of
// the LDS-DMA pieces (100-185 cycles each; the global -> LDS path delivers ~13-17 bytes per clock and CU whatever the schedule) and by
// the activation split, which every wave repeats for the rows it shares with its column neighbours -- not by the matrix pipe.  This
// member takes the other road:
//   * operands come through REGISTERS: per k-step a thread loads 8 fp32 activations (its row's two 16-byte chunks of one
//     (sub-step, lane-half) fragment: buffer_load_dwordx4 x 2) and three 16-byte weight chunks (one per plane), a step ahead;
//   * the activations are split into the three bf16 planes ONCE per element (44 VALU per thread and step instead of 88) and written to
//     LDS as planes (ds_write_b128 x 3, rows of 64 bytes, the chunk XOR-swizzled by the row: fragment reads and stage writes are
//     bank-conflict free); the weight chunks go to the image the DMA kernels use ([plane][chunk][column] x 16 bytes);
//   * the A fragments go from LDS straight into the MFMAs; two LDS buffers of 24 KB, ONE barrier per k-step, no inline-asm waits
//     (nothing here is an LDS-DMA, so hipcc's own counted waits are right);
//   * 48 KB of LDS and <= 168 registers: THREE workgroups per CU (the DMA kernel: two).
// Same k -> (sub-step, lane-half, element) mapping and the same six products in the same order per accumulator as gemm_x6_kernel:
// bit-identical results.
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
// WM = waves along M (2: 64-row tile, 4: 128-row tile), NBW = 32-column blocks per wave (two waves along N): <2, 1> = 64x64 on four
// waves, three workgroups per CU; <4, 2> = 128x128 on eight waves (each 32 rows x 64 columns, as gemm_x6w_kernel), one workgroup per
// CU -- half the weight bytes per product, for the layers whose 128x128 tiles fill the chip.
template <bool IS1X1, int WM, int NBW>
__global__ void __launch_bounds__(128 * WM, WM == 2 ? 3 : 2) gemm_x6r_kernel(const ConvParams p, const X6Weight wq) {
  constexpr int NT = 128 * WM;                            // threads: WM x 2 waves
  constexpr int BM = 32 * WM, BN = 64 * NBW;
  constexpr int A_PLANE = BM * 64;                        // bytes: BM rows x four 16-byte chunks (32 bf16)
  constexpr int A_BYTES = 3 * A_PLANE, B_PIECE = BN * 16, B_BYTES = 12 * B_PIECE;
  constexpr int BUF = A_BYTES + B_BYTES;                  // 24 KB (64x64) / 48 KB (128x128)
  constexpr unsigned OOB = 0x80000000u;
  static_assert(NT == 4 * BM && 3 * NT == 12 * BN, "one A fragment and three weight chunks per thread and k-step");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int nbn = (p.Cout + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
  const int nk = p.K / BK;
  const int nitems = nbm * nbn;
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;          // XCD-aware item order, as in gemm_lds_kernel
  const int nwg_x = ((int)gridDim.x - xcd + 7) >> 3;
  const int q8 = nitems >> 3, r8 = nitems & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
  const int mine = li < chunk_n ? (chunk_n - li + nwg_x - 1) / nwg_x : 0;
  const int total = mine * nk;
  if (total == 0) return;
  auto item_of = [&](int i) __attribute__((always_inline)) {
    const int it = chunk0 + li + i * nwg_x;
    Item r;
    r.bn = it % nbn;
    r.bm = it / nbn;
    r.kt0 = 0;
    return r;
  };
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32 * NBW;
  const int hw_out = p.OH * p.OW;
  const int plane_bytes = (p.K / 8) * wq.cout_pad * 16;
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)((long)p.B * p.H * p.W * p.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wq.w6), 0, 3 * plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_out =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((long)p.M * p.ldc * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_res =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, p.res ? (int)((long)(p.res_rows ? p.res_rows : p.M) * p.ldr * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_bias =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
  const bool has_res = p.res != nullptr, has_bias = p.bias != nullptr;

  // ---- staging side: thread = (row tid >> 2 of the tile, fragment slot sh = 2 s + h); weight chunk (cc = tid / BN, column tid % BN) ----
  const int srow = tid >> 2, sh = tid & 3;
  const int c0 = 4 * (sh >> 1) + (sh & 1);                       // its first 16-byte chunk of the row's 128 bytes; the second is c0 + 2
  const unsigned a_wr = (unsigned)(srow * 64 + ((sh ^ ((srow >> 2) & 3)) << 4));
  const int bcc = wave / (BN / 64), bcol = (wave % (BN / 64)) * 64 + lane;       // (the chunk column is wave-uniform)
  const unsigned b_wr = (unsigned)(A_BYTES + bcc * B_PIECE + bcol * 16);            // + pl * 4 pieces
  int is_i = 0, is_kt = 0;
  int a_off = 0, a_iy0 = 0, a_ix0 = 0;
  bool a_ok = false;
  unsigned b_off = 0;
  int s_k = 0, s_kb = 0;
  int tap_c = 0, tap_ky = 0, tap_kx = 0;
  u32x4v sa[2], sb[3];                                           // the staged step: 8 fp32 activations, three weight chunks
  auto setup_item = [&](int i) __attribute__((always_inline)) {
    const bool live = i < mine;
    const Item it = item_of(live ? i : 0);
    const int m = it.bm * BM + srow;
    a_ok = live && m < p.M;
    const int mm = a_ok ? m : 0;
    const int b = mm / hw_out, pix = mm - b * hw_out;
    const int oy = pix / p.OW, ox = pix - oy * p.OW;
    a_iy0 = oy * p.stride - p.pad;
    a_ix0 = ox * p.stride - p.pad;
    a_off = (((b * p.H + a_iy0) * p.W + a_ix0) * p.lda + 4 * c0) * 4;
    if (IS1X1 && !a_ok) a_off = (int)OOB;
    // (a 128-wide tile on a weight padded to 64 columns: the columns past the padded width are masked)
    b_off = (live && it.bn * BN + bcol < wq.cout_pad) ? (unsigned)((bcc * wq.cout_pad + it.bn * BN + bcol) * 16) : OOB;
    s_k = 0;
    s_kb = 0;
    if (!IS1X1) { tap_c = 0; tap_ky = 0; tap_kx = 0; }
  };
  auto gload = [&]() __attribute__((always_inline)) {            // global -> registers, the next step not yet staged
    if (is_kt == 0) setup_item(is_i);
    int voff = a_off;
    if (!IS1X1) {
      const int iy = a_iy0 + tap_ky * p.dil, ix = a_ix0 + tap_kx * p.dil;
      const bool in = a_ok & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      voff = in ? a_off + ((tap_ky * p.dil * p.W + tap_kx * p.dil) * p.lda + tap_c) * 4 : (int)OOB;
    }
    sa[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff, IS1X1 ? s_k : 0, 0);
    sa[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff + 32, IS1X1 ? s_k : 0, 0);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) sb[pl] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)b_off, s_kb + pl * plane_bytes, 0);
    s_k += BK * 4;
    s_kb += 4 * wq.cout_pad * 16;
    if (!IS1X1) {
      tap_c += BK;
      if (tap_c >= p.Cin) { tap_c = 0; if (++tap_kx == p.KW) { tap_kx = 0; ++tap_ky; } }
    }
    if (++is_kt == nk) { is_kt = 0; ++is_i; }
  };
  auto stage_write = [&](auto BUFI) __attribute__((always_inline)) {   // registers -> split -> LDS buffer BUFI
    unsigned char* st = lds + decltype(BUFI)::value * BUF;
    bf16x8 pl3[3];
    split3(__builtin_bit_cast(f32x4, sa[0]), __builtin_bit_cast(f32x4, sa[1]), pl3);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(st + pl * A_PLANE + a_wr) = pl3[pl];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4v*>(st + b_wr + pl * 4 * B_PIECE) = sb[pl];
  };

  // ---- compute side ------------------------------------------------------------------------------------------------
  const int frow = wm + l31;
  unsigned a_rd[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) a_rd[s] = (unsigned)(frow * 64 + (((2 * s + half) ^ ((frow >> 2) & 3)) << 4));
  const unsigned b_rd = (unsigned)(A_BYTES + half * B_PIECE + (wn + l31) * 16);     // chunk column cc = 2 s + half: + 2 s pieces; block nb: + 512
  f32x16 acc[2][NBW];          // [sub-step][column block]
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][nb][r] = 0.f;
  int c_i = 0, c_kt = 0;
  float rv[NBW][16], bv[NBW];
  auto epi_loads = [&]() __attribute__((always_inline)) {        // residual and bias of the tile, under its last k-step
    const Item it = item_of(c_i);
    const int m0 = it.bm * BM;
    const int rr0 = p.res_rows ? m0 % p.res_rows : m0;            // (scalar: once per tile)
    const bool wrap1 = p.res_rows >= BM;                          // a shared map at least a tile tall: at most one wrap
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int n = it.bn * BN + wn + 32 * nb + l31;
      const bool col_ok = n < p.Cout;
      if (has_bias) bv[nb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_bias, col_ok ? n * 4 : (int)OOB, 0, 0));
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dm = wm + mfma32_row(r, half);
          int rr = rr0 + dm;
          if (p.res_rows) {
            if (wrap1) rr = rr >= p.res_rows ? rr - p.res_rows : rr;
            else rr %= p.res_rows;
          }
          rv[nb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                    rsrc_res, (col_ok && m0 + dm < p.M) ? (rr * p.ldr + n) * 4 : (int)OOB, 0, 0));
        }
      }
    }
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    const Item it = item_of(c_i);
    const int mlane = it.bm * BM + wm + 4 * half;
    const int rows_left = p.M - mlane, ldc4 = p.ldc * 4;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int n = it.bn * BN + wn + 32 * nb + l31;
      const bool col_ok = n < p.Cout;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][nb][r] += acc[1][nb][r]; acc[1][nb][r] = 0.f; }
      const int vbase = col_ok ? (mlane * p.ldc + n) * 4 : (int)OOB;
      if (has_bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nb][r] += bv[nb];
      }
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nb][r] += rv[nb][r];
      }
      with_act(p.act, [&](auto ACT) __attribute__((always_inline)) -> void {
        constexpr int act = decltype(ACT)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, apply_act(acc[0][nb][r], act)), rsrc_out,
                                                c < rows_left ? vbase : (int)OOB, c * ldc4, 0);
          acc[0][nb][r] = 0.f;
        }
      });
    }
  };
  auto wg_barrier = [&]() __attribute__((always_inline)) {       // LDS writes of this wave done, then everybody's
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto step = [&](auto BUFI) __attribute__((always_inline)) -> void {
    constexpr int bi = decltype(BUFI)::value;
    const unsigned char* st = lds + bi * BUF;
    bf16x8 fa[3][2], fb[3][2][NBW];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        fa[pl][s] = *reinterpret_cast<const bf16x8*>(st + pl * A_PLANE + a_rd[s]);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
          fb[pl][s][nb] = *reinterpret_cast<const bf16x8*>(st + b_rd + (pl * 4 + 2 * s) * B_PIECE + nb * 512);
      }
    if (c_kt == nk - 1 && (has_res | has_bias)) epi_loads();
    // smallest terms first per accumulator; sub-steps and column blocks alternate (consecutive MFMAs are independent).  The staging
    // of the NEXT step (split + LDS writes) and the global loads of the one after sit between the two halves of the MFMA chain.
#define AOT_X6R_TERM(PA, PB)                                                                                  \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)             \
      acc[s][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][s], fb[PB][s][nb], acc[s][nb], 0, 0, 0);
    AOT_X6R_TERM(1, 1)
    AOT_X6R_TERM(0, 2)
    AOT_X6R_TERM(2, 0)
    stage_write(std::integral_constant<int, bi ^ 1>{});          // the step after this one: registers -> the other buffer
    gload();                                                     // the step after that: global -> registers
    AOT_X6R_TERM(0, 1)
    AOT_X6R_TERM(1, 0)
    AOT_X6R_TERM(0, 0)
#undef AOT_X6R_TERM
    if (++c_kt == nk) {
      epilogue();
      c_kt = 0;
      ++c_i;
    }
    wg_barrier();
  };
  gload();
  stage_write(std::integral_constant<int, 0>{});
  gload();
  wg_barrier();
#pragma unroll 1
  for (int ss = 0; ss < total; ss += 2) {
    step(std::integral_constant<int, 0>{});
    if (ss + 1 < total) step(std::integral_constant<int, 1>{});
  }
}

// The 64x64 form with the WEIGHT fragments taken straight from global memory ("x6rd", round 5).  SQ counters of gemm_x6r_kernel<., 2, 1>
// (profiles/r05_x6_gemm_pmc.txt): the LDS is as busy as the matrix pipe -- per k-step a workgroup writes 24 KB (ds_write_b128 moves
// ~79 bytes per clock) and reads 48 KB -- and half of both is the weight tile, which needs no transposition at all: the packed planes
// (aot_pack_bf16x6_f32) ARE the fragments (lane = column, 16 bytes = the eight k of a lane-half and sub-step), one 512-byte run per lane
// half.  So every wave loads its six weight fragments of the NEXT step into a second register set (buffer_load_dwordx4 x 6; the two
// row waves of a column fetch the same lines, the second one from the CU's L1) and only the activation planes go through the LDS:
// 12 KB written + 24 KB read per step, two buffers of 12 KB.  Same products in the same order: bit-identical to the other 64x64 forms.
// SK: split-K over the grid (item = (tile, k-slice); raw partial tiles to fp32 slabs [ksplit][M][Cout], summed in slice order by
// splitk_reduce_kernel): the long-K 3x3 layers on the stride-16 map have 108-316 tiles of 72 k-steps each -- too few workgroups,
// too long a chain.
// GN (not with SK; `scratch` then carries the partial-sum buffer): the tile end also writes the GroupNorm partial sums of its output --
// every wave owns a 32-row x 32-column block, i.e. 32 rows of ONE 32-channel group: (sum, sum of squared deviations from the block's
// own mean) of the block's valid elements (fp32, the stored values themselves) -> gn_part[(2 * tile row + wave row) * (Cout / 32) + column block][2].  The consumer
// (gn_act_dwconv5_kernel<true>) adds the partials of a group in index order in double: the statistics pass over the whole map and its
// launch are gone (linear1 -> GN -> GELU -> dw5x5 of the LSTT's feed-forward, transformer.py:355-362 / basic.py:15-35).
// C4 (the ResNet stem, 7x7 stride 2 on the image padded to FOUR channels): Cin = 4 makes one 16-byte chunk of the A row exactly one
// filter tap (r, g, b, 0 of one input pixel), so a k-step is eight taps instead of 32 channels of one tap: the thread's two chunks are
// two taps with a bounds check each; K = KH * KW * 4 rounded up to 32 (the weight rows past it are zero: aot_pack_bf16x6_f32 of the
// zero-padded matrix).  The last big layer that was still on the fp32 matrix cores in bf16x6 engines.
// LN (a linear layer, unsplit; round 6 -- the `aot_layernorm_linear` of SURVEY 8b, transformer.py:321-359): the A operand is the
// LayerNorm of `in` over its K channels, never materialised.  gamma is folded into the weight and beta into the bias by the host
// (W' = diag(gamma) W, b' = beta W + b), so the kernel owes (x - mean) * rstd per row -- and both statistics RIDE ALONG the k-loop, no
// pass over the rows in front of it (a first version read the rows once for the mean before the pipeline started: as slow as the
// LayerNorm launch it replaced, profiles/r06_fusions_ab.txt).  With c = the row's first element (a shift inside the row's range):
// d = x - c is the operand that gets split into planes; sum d and sum d^2 are added up by the staging threads on the way (four threads
// per row, quad reduce after the item's last step); then m' = mean - c = sum d / K, var = sum d^2 / K - m'^2 (|m'| is of the order of
// the row's spread: no cancellation between large numbers, neither here nor in the product), and since
//     (x - mean) W' = d W' - m' * colsum(W')
// the tile end turns the accumulator d W' into rstd * (acc - m' * s_n) with s = the column sums of W' (ln_colsum, from the host).
// (m', rstd) reach the tile end through 2 x 64 floats of LDS, double-buffered by item parity: the staging side runs two steps ahead.
// GROUP (round 6): up to four products of ONE shape in one launch -- blockIdx.y names the problem, whose operand pointers replace those
// of `p` / `wq` (independent linear layers on a stride-16 map fill 108 of 256 CUs each: the three layers' linear_V of the memory
// update, the four value / gate projections of a GPM block's self-propagation).
struct X6Group {
  const float* in[4];
  const void* w6[4];
  const float* bias[4];
  const float* res[4];
  float* out[4];
};
template <bool IS1X1, bool SK, bool GN = false, bool C4 = false, bool LN = false, bool GROUP = false>
__global__ void __launch_bounds__(256, 3) gemm_x6rd_kernel(const ConvParams p_in, const X6Weight wq_in, const int ksplit, float* __restrict__ scratch,
                                                           const float ln_eps, const float* __restrict__ ln_colsum, const X6Group grp) {
  ConvParams p = p_in;
  X6Weight wq = wq_in;
  if (GROUP) {
    const int g = blockIdx.y;
    p.in = grp.in[g]; p.bias = grp.bias[g]; p.res = grp.res[g]; p.out = grp.out[g];
    wq.w6 = grp.w6[g];
  }
  static_assert(!(SK && GN), "GroupNorm partials come from the unsplit form");
  static_assert(!C4 || (!IS1X1 && !SK && !GN), "the four-channel form: a KxK layer, unsplit");
  static_assert(!LN || (IS1X1 && !SK && !C4), "the LayerNorm prologue: a linear layer, unsplit");
  constexpr int WM = 2, NBW = 1;
  constexpr int NT = 128 * WM;                            // threads: WM x 2 waves
  constexpr int BM = 32 * WM, BN = 64 * NBW;
  constexpr int A_PLANE = BM * 64;                        // bytes: BM rows x four 16-byte chunks (32 bf16)
  constexpr int A_BYTES = 3 * A_PLANE;
  constexpr int BUF = A_BYTES;                            // 12 KB: the activation planes only
  constexpr unsigned OOB = 0x80000000u;
  static_assert(NT == 4 * BM, "one A fragment per thread and k-step");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
  __shared__ __attribute__((aligned(16))) float ln_rstd[LN ? 2 * BM : 4];       // LN: 1 / sqrt(var + eps) of the tile's rows, by item parity
  __shared__ __attribute__((aligned(16))) float ln_mp[LN ? 2 * BM : 4];         // LN: mean - shift of the tile's rows

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int nbn = (p.Cout + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
  const int nk = C4 ? (p.KH * p.KW + 7) / 8 : SK ? (p.K / BK) / ksplit : p.K / BK;      // k-steps per item (host guarantees divisibility)
  const int nitems = nbm * nbn * (SK ? ksplit : 1);
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;          // XCD-aware item order, as in gemm_lds_kernel
  const int nwg_x = ((int)gridDim.x - xcd + 7) >> 3;
  const int q8 = nitems >> 3, r8 = nitems & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
  const int mine = li < chunk_n ? (chunk_n - li + nwg_x - 1) / nwg_x : 0;
  const int total = mine * nk;
  if (total == 0) return;
  auto item_of = [&](int i) __attribute__((always_inline)) {
    const int it = chunk0 + li + i * nwg_x;
    Item r;
    r.bn = it % nbn;
    if (SK) {            // item = ((bm * ksplit) + slice) * nbn + bn: the slices of a tile are neighbours
      const int t = it / nbn;
      r.kt0 = (t % ksplit) * nk;
      r.bm = t / ksplit;
    } else {
      r.bm = it / nbn;
      r.kt0 = 0;
    }
    return r;
  };
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32 * NBW;
  const int hw_out = p.OH * p.OW;
  const int plane_bytes = (C4 ? nk * 4 : p.K / 8) * wq.cout_pad * 16;
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)((long)p.B * p.H * p.W * p.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wq.w6), 0, 3 * plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_out =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((long)p.M * p.ldc * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_res =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, p.res ? (int)((long)(p.res_rows ? p.res_rows : p.M) * p.ldr * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_bias =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
  const bool has_res = !SK && p.res != nullptr, has_bias = !SK && p.bias != nullptr;      // (split-K: the reduce pass adds them)

  // ---- staging side: thread = (row tid >> 2 of the tile, fragment slot sh = 2 s + h); weight chunk (cc = tid / BN, column tid % BN) ----
  const int srow = tid >> 2, sh = tid & 3;
  const int c0 = 4 * (sh >> 1) + (sh & 1);                       // its first 16-byte chunk of the row's 128 bytes; the second is c0 + 2
  const unsigned a_wr = (unsigned)(srow * 64 + ((sh ^ ((srow >> 2) & 3)) << 4));
  int is_i = 0, is_kt = 0;
  int a_off = 0, a_iy0 = 0, a_ix0 = 0;
  bool a_ok = false;
  int s_k = 0;
  // weight side: its own walk over the items, ONE step ahead of the MFMAs (the activations are two steps ahead: one in registers,
  // one in LDS); lane (column wn + l31, half) fetches chunk column cc = 2 s + half of plane pl: + (pl planes, 2 s chunk columns) scalar
  int ib_i = 0, ib_kt = 0, s_kb = 0;
  unsigned b_off = 0;
  int tap_c = 0, tap_ky = 0, tap_kx = 0;
  u32x4v sa[2];                                                  // the staged step: 8 fp32 activations
  bf16x8 fb[2][3][2];                                            // [register set][plane][sub-step]: this step's and the next step's weights
  float ln_c = 0.f, ln_s = 0.f, ln_q = 0.f;                      // LN: the row's shift; this thread's share of sum (x - c) and sum (x - c)^2
  int sw_kt = 0, sw_i = 0;                                       // LN: the staging side's own step / item counters
  auto setup_item = [&](int i) __attribute__((always_inline)) {
    const bool live = i < mine;
    const Item it = item_of(live ? i : 0);
    const int m = it.bm * BM + srow;
    a_ok = live && m < p.M;
    const int mm = a_ok ? m : 0;
    const int b = mm / hw_out, pix = mm - b * hw_out;
    const int oy = pix / p.OW, ox = pix - oy * p.OW;
    a_iy0 = oy * p.stride - p.pad;
    a_ix0 = ox * p.stride - p.pad;
    a_off = (((b * p.H + a_iy0) * p.W + a_ix0) * p.lda + (C4 ? 0 : 4 * c0)) * 4;
    if (IS1X1 && !a_ok) a_off = (int)OOB;
    s_k = SK ? it.kt0 * BK * 4 : 0;
    if (!IS1X1) {
      if (SK) {            // the slice's first k-step names its filter tap
        const int k0 = it.kt0 * BK, tap = k0 / p.Cin;
        tap_c = k0 - tap * p.Cin;
        tap_ky = tap / p.KW;
        tap_kx = tap - tap_ky * p.KW;
      } else {
        tap_c = 0; tap_ky = 0; tap_kx = 0;
      }
    }
  };
  auto gload = [&]() __attribute__((always_inline)) {            // global -> registers, the next step not yet staged
    if (is_kt == 0) setup_item(is_i);
    if (C4) {            // chunk = tap: taps 8 is_kt + c0 and + 2 of the filter, each inside the image or not
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int tap = 8 * is_kt + c0 + 2 * j;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int iy = a_iy0 + ky * p.dil, ix = a_ix0 + kx * p.dil;
        const bool in = a_ok & (tap < p.KH * p.KW) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        sa[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, in ? a_off + ((ky * p.dil * p.W + kx * p.dil) * p.lda) * 4 : (int)OOB, 0, 0);
      }
      if (++is_kt == nk) { is_kt = 0; ++is_i; }
      return;
    }
    int voff = a_off;
    if (!IS1X1) {
      const int iy = a_iy0 + tap_ky * p.dil, ix = a_ix0 + tap_kx * p.dil;
      const bool in = a_ok & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      voff = in ? a_off + ((tap_ky * p.dil * p.W + tap_kx * p.dil) * p.lda + tap_c) * 4 : (int)OOB;
    }
    sa[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff, IS1X1 ? s_k : 0, 0);
    sa[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff + 32, IS1X1 ? s_k : 0, 0);
    s_k += BK * 4;
    if (!IS1X1) {
      tap_c += BK;
      if (tap_c >= p.Cin) { tap_c = 0; if (++tap_kx == p.KW) { tap_kx = 0; ++tap_ky; } }
    }
    if (++is_kt == nk) { is_kt = 0; ++is_i; }
  };
  auto gload_b = [&](auto SET) __attribute__((always_inline)) {       // the weight fragments of the next step -> register set SET
    constexpr int q = decltype(SET)::value;
    if (ib_kt == 0) {
      const bool live = ib_i < mine;
      const Item it = item_of(live ? ib_i : 0);
      b_off = live ? (unsigned)((half * wq.cout_pad + it.bn * BN + wn + l31) * 16) : OOB;
      s_kb = SK ? it.kt0 * 4 * wq.cout_pad * 16 : 0;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        fb[q][pl][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)b_off,
                                                                                         s_kb + pl * plane_bytes + 2 * s * wq.cout_pad * 16, 0));
    s_kb += 4 * wq.cout_pad * 16;
    if (++ib_kt == nk) { ib_kt = 0; ++ib_i; }
  };
  auto stage_write = [&](auto BUFI) __attribute__((always_inline)) {   // registers -> split -> LDS buffer BUFI
    unsigned char* st = lds + decltype(BUFI)::value * BUF;
    bf16x8 pl3[3];
    if (LN) {            // d = x - c is the operand; sum d and sum d^2 give the row's mean and variance at the item's end
      f32x4 d0 = __builtin_bit_cast(f32x4, sa[0]), d1 = __builtin_bit_cast(f32x4, sa[1]);
      if (sw_kt == 0) ln_c = __shfl(d0[0], lane & ~3);            // the row's first element (the quad's slot-0 thread holds chunk 0)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d0[e] -= ln_c;
        d1[e] -= ln_c;
        ln_s += d0[e] + d1[e];
        ln_q = fmaf(d0[e], d0[e], ln_q);
        ln_q = fmaf(d1[e], d1[e], ln_q);
      }
      split3(d0, d1, pl3);
      if (++sw_kt == nk) {            // the item's last step is staged: (m', rstd) of the row -> LDS (read by the tile end >= one barrier later)
        float sd = ln_s + __shfl_xor(ln_s, 1), q = ln_q + __shfl_xor(ln_q, 1);
        sd += __shfl_xor(sd, 2);
        q += __shfl_xor(q, 2);
        const float mp = sd / (float)p.K;
        const float var = fmaxf(q / (float)p.K - mp * mp, 0.f);
        if (sh == 0) {
          ln_rstd[(sw_i & 1) * BM + srow] = 1.f / sqrtf(var + ln_eps);
          ln_mp[(sw_i & 1) * BM + srow] = mp;
        }
        ln_s = 0.f;
        ln_q = 0.f;
        sw_kt = 0;
        ++sw_i;
      }
    } else {
      split3(__builtin_bit_cast(f32x4, sa[0]), __builtin_bit_cast(f32x4, sa[1]), pl3);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(st + pl * A_PLANE + a_wr) = pl3[pl];
  };

  // ---- compute side ------------------------------------------------------------------------------------------------
  const int frow = wm + l31;
  unsigned a_rd[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) a_rd[s] = (unsigned)(frow * 64 + (((2 * s + half) ^ ((frow >> 2) & 3)) << 4));
  f32x16 acc[2][NBW];          // [sub-step][column block]
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][nb][r] = 0.f;
  int c_i = 0, c_kt = 0;
  float rv[NBW][16], bv[NBW];
  auto epi_loads = [&]() __attribute__((always_inline)) {        // residual and bias of the tile, under its last k-step
    const Item it = item_of(c_i);
    const int m0 = it.bm * BM;
    const int rr0 = p.res_rows ? m0 % p.res_rows : m0;            // (scalar: once per tile)
    const bool wrap1 = p.res_rows >= BM;                          // a shared map at least a tile tall: at most one wrap
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int n = it.bn * BN + wn + 32 * nb + l31;
      const bool col_ok = n < p.Cout;
      if (has_bias) bv[nb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_bias, col_ok ? n * 4 : (int)OOB, 0, 0));
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dm = wm + mfma32_row(r, half);
          int rr = rr0 + dm;
          if (p.res_rows) {
            if (wrap1) rr = rr >= p.res_rows ? rr - p.res_rows : rr;
            else rr %= p.res_rows;
          }
          rv[nb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                    rsrc_res, (col_ok && m0 + dm < p.M) ? (rr * p.ldr + n) * 4 : (int)OOB, 0, 0));
        }
      }
    }
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    const Item it = item_of(c_i);
    const int mlane = it.bm * BM + wm + 4 * half;
    const int rows_left = p.M - mlane, ldc4 = p.ldc * 4;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int n = it.bn * BN + wn + 32 * nb + l31;
      const bool col_ok = n < p.Cout;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][nb][r] += acc[1][nb][r]; acc[1][nb][r] = 0.f; }
      if (SK) {          // the raw partial tile -> the slice's slab [M][Cout]
        const __amdgpu_buffer_rsrc_t rsrc_slab = __builtin_amdgcn_make_buffer_rsrc(
            scratch + (long)(it.kt0 / nk) * p.M * p.Cout, 0, (int)((long)p.M * p.Cout * 4), 0x00020000);
        const int vbase_s = col_ok ? (mlane * p.Cout + n) * 4 : (int)OOB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const float v = acc[0][nb][r];       // (a float of its own: __builtin_bit_cast applied to the vector ELEMENT expression
                                               //  made hipcc store zeros for every element but the first -- seen in the ISA)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_slab,
                                                c < rows_left ? vbase_s : (int)OOB, c * p.Cout * 4, 0);
          acc[0][nb][r] = 0.f;
        }
        continue;
      }
      const int vbase = col_ok ? (mlane * p.ldc + n) * 4 : (int)OOB;
      if (LN) {            // d W' is in the accumulator: (x - mean) W' * rstd = (acc - m' * colsum) * rstd
        const float* rs = ln_rstd + (c_i & 1) * BM + wm + 4 * half;
        const float* ms = ln_mp + (c_i & 1) * BM + wm + 4 * half;
        const float sn = col_ok ? ln_colsum[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2);
          acc[0][nb][r] = (acc[0][nb][r] - ms[rr] * sn) * rs[rr];
        }
      }
      if (has_bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nb][r] += bv[nb];
      }
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nb][r] += rv[nb][r];
      }
      float ps = 0.f;
      float gv[16];
      with_act(p.act, [&](auto ACT) __attribute__((always_inline)) -> void {
        constexpr int act = decltype(ACT)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const float v = apply_act(acc[0][nb][r], act);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_out, c < rows_left ? vbase : (int)OOB, c * ldc4, 0);
          if (GN) {
            gv[r] = v;
            if (c < rows_left && col_ok) ps += v;
          }
          acc[0][nb][r] = 0.f;
        }
      });
      // the wave's 32 x 32 block = 32 rows of one group (the column block lies inside Cout: a block past it -- Cout % 64 == 32 -- writes
      // nothing).  Partial = (sum, sum of squared deviations from the BLOCK's own mean): the consumer combines the blocks
      // with Chan's formula -- no E[x^2] - mean^2 cancellation when |mean| >> std.  Fixed butterflies over the 64 lanes.
      if (GN && it.bn * BN + wn + 32 * nb < p.Cout) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ps += __shfl_xor(ps, off);
        const int brows = min(32, max(0, p.M - (it.bm * BM + wm)));
        const float mb = brows > 0 ? ps / (float)(brows * 32) : 0.f;
        float pq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const float dv = gv[r] - mb;
          if (c < rows_left && col_ok) pq += dv * dv;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pq += __shfl_xor(pq, off);
        if (lane == 0) {
          float* dst = scratch + ((long)(it.bm * 2 + (wave >> 1)) * (p.Cout >> 5) + ((it.bn * BN + wn + 32 * nb) >> 5)) * 2;
          dst[0] = ps;
          dst[1] = pq;
        }
      }
    }
  };
  auto wg_barrier = [&]() __attribute__((always_inline)) {       // LDS writes of this wave done, then everybody's
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto step = [&](auto BUFI) __attribute__((always_inline)) -> void {
    constexpr int bi = decltype(BUFI)::value;
    const unsigned char* st = lds + bi * BUF;
    bf16x8 fa[3][2];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int s = 0; s < 2; ++s) fa[pl][s] = *reinterpret_cast<const bf16x8*>(st + pl * A_PLANE + a_rd[s]);
    if (c_kt == nk - 1 && (has_res | has_bias)) epi_loads();
    // smallest terms first per accumulator; sub-steps and column blocks alternate (consecutive MFMAs are independent).  The staging
    // of the NEXT step (split + LDS writes) and the global loads of the one after sit between the two halves of the MFMA chain.
#define AOT_X6R_TERM(PA, PB)                                                                                  \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)             \
      acc[s][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][s], fb[bi][PB][s], acc[s][nb], 0, 0, 0);
    AOT_X6R_TERM(1, 1)
    AOT_X6R_TERM(0, 2)
    AOT_X6R_TERM(2, 0)
    stage_write(std::integral_constant<int, bi ^ 1>{});          // the step after this one: registers -> the other buffer
    gload();                                                     // the step after that: global -> registers
    gload_b(std::integral_constant<int, bi ^ 1>{});              // the next step's weight fragments -> the other register set
    AOT_X6R_TERM(0, 1)
    AOT_X6R_TERM(1, 0)
    AOT_X6R_TERM(0, 0)
#undef AOT_X6R_TERM
    if (++c_kt == nk) {
      epilogue();
      c_kt = 0;
      ++c_i;
    }
    wg_barrier();
  };
  gload();
  stage_write(std::integral_constant<int, 0>{});
  gload();
  gload_b(std::integral_constant<int, 0>{});
  wg_barrier();
#pragma unroll 1
  for (int ss = 0; ss < total; ss += 2) {
    step(std::integral_constant<int, 0>{});
    if (ss + 1 < total) step(std::integral_constant<int, 1>{});
  }
}

// fragment reads of the 128x128 LDS-DMA tile (gemm_x6pp_kernel): A fragments double-buffered, weight fragments of the current step
// (planes x sub-steps x column blocks; three stages of 41 KB: the stage offset does not fit the 16-bit immediate, it is added to the
// address registers)
__device__ __forceinline__ void x6w_fetch_a(f32x4 (&a)[4], const unsigned (&aaddr)[4], unsigned stage) {
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(a[j]) : "v"(aaddr[j] + stage));
}
template <int PIECE>
__device__ __forceinline__ void x6w_fetch_b(bf16x8 (&b)[3][2][2], unsigned baddr) {
#define AOT_X6W_B(PL, S, NB) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[PL][S][NB]) : "v"(baddr), "n"(((PL) * 4 + 2 * (S)) * 2 * PIECE + (NB) * 512));
  AOT_X6W_B(0, 0, 0) AOT_X6W_B(0, 0, 1) AOT_X6W_B(0, 1, 0) AOT_X6W_B(0, 1, 1)
  AOT_X6W_B(1, 0, 0) AOT_X6W_B(1, 0, 1) AOT_X6W_B(1, 1, 0) AOT_X6W_B(1, 1, 1)
  AOT_X6W_B(2, 0, 0) AOT_X6W_B(2, 0, 1) AOT_X6W_B(2, 1, 0) AOT_X6W_B(2, 1, 1)
#undef AOT_X6W_B
}
__device__ __forceinline__ void x6w_landed_a(f32x4 (&a)[4]) { asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }
__device__ __forceinline__ void x6w_landed_b(bf16x8 (&b)[3][2][2]) {
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(b[pl][0][0]), "+v"(b[pl][0][1]), "+v"(b[pl][1][0]), "+v"(b[pl][1][1]));
}

// ---- the 128x128 tile in two PHASE-SHIFTED wave groups ("ping-pong"; round 5) ------------------------------------------------
// gemm_x6w_kernel's eight waves all walk the same sequence inside a k-step -- weight fragments, split, MFMAs -- so the matrix pipe
// idles while every wave reads and splits, and the vector pipe idles while every wave multiplies: the steady state of that kernel
// is ~3100 cycles per k-step against 1536 of MFMA per SIMD (dec c4 at batch 3: 203 TF-equivalent once tile quantisation is taken
// out).  Here the two waves that share a SIMD (w and w + 4: a workgroup's waves go to the SIMDs cyclically) work in OPPOSITE phases,
// separated by workgroup barriers:
//     group X (waves 0-3):   LOAD(s) | COMPUTE(s) | LOAD(s+1) | COMPUTE(s+1) | ...
//     group Y (waves 4-7):     --    | LOAD(s)    | COMPUTE(s)| LOAD(s+1)    | ...
// LOAD(s) = the wave's 4 + 12 fragment reads of ring stage s, the split of its A rows into the three bf16 planes (88 VALU), and the
// issue of its five LDS-DMA pieces of step s+2; COMPUTE(s) = its 24 MFMAs (768 cycles), all operands in registers.  In every phase
// one wave of a SIMD feeds the matrix pipe while its partner uses the LDS and the vector ALUs.  Same tile, LDS image, DMA pieces,
// six products in the same order per accumulator and tile end as gemm_x6w_kernel: bit-identical results.
// Ring safety (three stages): stage s is read by X in phase 2s and by Y in phase 2s+1; the DMA of step s+2 goes to the stage of
// step s-1, whose last read (Y, phase 2s-1) is behind a barrier for both groups.  Every wave retires its own pieces of step s+1
// (counted vmcnt) before the barrier that ends phase 2s+1: X at the end of COMPUTE(s), Y at the end of LOAD(s).
// SK: split-K over the grid (item = (tile, k-slice)), raw partial tiles to fp32 slabs, splitk_reduce_kernel sums them in slice
// order -- for the stride-16 maps, whose 128x128 tiles alone do not fill 256 CUs.
template <bool IS1X1, bool SK>
__global__ void __launch_bounds__(512, 2) gemm_x6pp_kernel(const ConvParams p, const X6Weight wq, const int ksplit, float* __restrict__ scratch) {
  constexpr int NST = 3;
  constexpr int BM = 128, BN = 128;
  constexpr int AG = BM / 8, AGW = AG / 8;              // A: 8-row groups, two per wave (eight waves)
  constexpr int BPW = 3;                                // B: one 16-byte chunk column (cc = wave & 3) of each plane per wave
  constexpr int LPW = AGW + BPW;
  constexpr int OPA_BYTES = AG * GROUP_STRIDE, B_PIECE = 64 * 16, OPB_BYTES = 24 * B_PIECE;      // piece (pl, cc, column half)
  constexpr int STAGE_BYTES = OPA_BYTES + OPB_BYTES;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                                     // 0 = X, 1 = Y (one phase behind)
  const int l31 = lane & 31, half = lane >> 5;
  const int nbn = (p.Cout + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
  const int nk = SK ? (p.K / BK) / ksplit : p.K / BK;            // k-steps per item (host guarantees divisibility)
  const int nitems = nbm * nbn * (SK ? ksplit : 1);
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;          // XCD-aware item order, as in gemm_lds_kernel
  const int nwg_x = ((int)gridDim.x - xcd + 7) >> 3;
  const int q8 = nitems >> 3, r8 = nitems & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
  const int mine = li < chunk_n ? (chunk_n - li + nwg_x - 1) / nwg_x : 0;
  const int total = mine * nk;
  if (total == 0) return;
  auto item_of = [&](int i) __attribute__((always_inline)) {
    const int it = chunk0 + li + i * nwg_x;
    Item r;
    r.bn = it % nbn;
    if (SK) {            // item = ((bm * ksplit) + slice) * nbn + bn: the slices of a tile are neighbours (shared A rows in L2)
      const int t = it / nbn;
      r.kt0 = (t % ksplit) * nk;
      r.bm = t / ksplit;
    } else {
      r.bm = it / nbn;
      r.kt0 = 0;
    }
    return r;
  };
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;       // wave tile: 32 rows x 64 columns (two 32-column blocks)
  const int lr = lane >> 3, lp = lane & 7;
  const int cofs = (lp ^ lr) << 2;
  const int hw_out = p.OH * p.OW;
  const int plane_bytes = (p.K / 8) * wq.cout_pad * 16;
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)((long)p.B * p.H * p.W * p.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wq.w6), 0, 3 * plane_bytes, 0x00020000);
  const i32x4 desc_out = raw_desc(p.out, (long)p.M * p.ldc * 4);
  const i32x4 desc_res = raw_desc(p.res, (long)(p.res_rows ? p.res_rows : p.M) * p.ldr * 4);
  const i32x4 desc_bias = raw_desc(p.bias, (long)p.Cout * 4);
  const int n_res = (!SK && p.res) ? 32 : 0, n_bias = (!SK && p.bias) ? 1 : 0;      // (split-K: the reduce pass adds them)

  // ---- issue side (as gemm_x6w_kernel; split-K: the slice's first k-step sets the K offsets and the filter tap) --------
  int is_i = 0, is_kt = 0;
  int a_off[AGW], a_iy0[AGW], a_ix0[AGW];
  bool a_ok[AGW];
  unsigned b_off = 0;
  int s_k = 0, s_kb = 0;
  int tap_c = 0, tap_ky = 0, tap_kx = 0, s_tap = 0;
  auto setup_item = [&](int i) __attribute__((always_inline)) {
    const bool live = i < mine;
    const Item it = item_of(live ? i : 0);
#pragma unroll
    for (int g = 0; g < AGW; ++g) {
      const int m = it.bm * BM + 8 * (AGW * wave + g) + lr;
      a_ok[g] = live && m < p.M;
      const int mm = a_ok[g] ? m : 0;
      const int b = mm / hw_out, pix = mm - b * hw_out;
      const int oy = pix / p.OW, ox = pix - oy * p.OW;
      a_iy0[g] = oy * p.stride - p.pad;
      a_ix0[g] = ox * p.stride - p.pad;
      a_off[g] = (((b * p.H + a_iy0[g]) * p.W + a_ix0[g]) * p.lda + cofs) * 4;
      if (IS1X1 && !a_ok[g]) a_off[g] = (int)OOB;
    }
    b_off = (live && it.bn * BN + (wave >> 2) * 64 < wq.cout_pad)
                ? (unsigned)(((wave & 3) * wq.cout_pad + it.bn * BN + (wave >> 2) * 64 + lane) * 16) : OOB;
    s_k = SK ? it.kt0 * BK * 4 : 0;
    s_kb = SK ? it.kt0 * 4 * wq.cout_pad * 16 : 0;
    if (!IS1X1) {
      if (SK) {
        const int k0 = it.kt0 * BK, tap = k0 / p.Cin;
        tap_c = k0 - tap * p.Cin;
        tap_ky = tap / p.KW;
        tap_kx = tap - tap_ky * p.KW;
      } else {
        tap_c = 0; tap_ky = 0; tap_kx = 0;
      }
    }
  };
  auto issue = [&](auto SLOT) __attribute__((always_inline)) -> void {
    constexpr int slot = decltype(SLOT)::value;
    if (is_kt == 0) setup_item(is_i);
    if (!IS1X1) s_tap = ((tap_ky * p.dil * p.W + tap_kx * p.dil) * p.lda + tap_c) * 4;
    unsigned char* st = lds + slot * STAGE_BYTES;
#pragma unroll
    for (int g = 0; g < AGW; ++g) {
      unsigned char* dst = st + (AGW * wave + g) * GROUP_STRIDE;
      if (IS1X1) {
        dma16(rsrc_a, dst, a_off[g], s_k);
      } else {
        const int iy = a_iy0[g] + tap_ky * p.dil, ix = a_ix0[g] + tap_kx * p.dil;
        const bool in = a_ok[g] & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        dma16(rsrc_a, dst, in ? a_off[g] + s_tap : (int)OOB, 0);
      }
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      dma16(rsrc_b, st + OPA_BYTES + ((pl * 4 + (wave & 3)) * 2 + (wave >> 2)) * B_PIECE, (int)b_off, s_kb + pl * plane_bytes);
    s_k += BK * 4;
    s_kb += 4 * wq.cout_pad * 16;
    if (!IS1X1) {
      tap_c += BK;
      if (tap_c >= p.Cin) { tap_c = 0; if (++tap_kx == p.KW) { tap_kx = 0; ++tap_ky; } }
    }
    if (++is_kt == nk) { is_kt = 0; ++is_i; }
  };

  // ---- compute side ------------------------------------------------------------------------------------------------
  const unsigned lds_base = (unsigned)(size_t)(lptr_t)lds;
  unsigned aaddr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) aaddr[j] = lds_base + chunk_off(wm + l31, 2 * j + half);
  const unsigned baddr = lds_base + OPA_BYTES + (half * 2 + (wn >> 6)) * B_PIECE + l31 * 16;
  f32x4 ra[4];                 // the lane's four 16-byte chunks of its A row: sub-step s contracts chunks 2 s and 2 s + 1
  bf16x8 rb[3][2][2];          // [plane][sub-step][column block]
  bf16x8 ap[2][3];             // [sub-step][plane]: the A row split, ready for the matrix cores
  f32x16 acc[4];               // [2 * sub-step + column block]
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  int c_i = 0, c_kt = 0;
  bool stores_pending = false;
  float rv[2][16], bv[2] = {0.f, 0.f};
  auto epi_loads = [&]() __attribute__((always_inline)) {
    const Item it = item_of(c_i);
    const int m0 = it.bm * BM;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = it.bn * BN + wn + 32 * nb + l31;
      const bool col_ok = n < p.Cout;
      if (n_bias) bv[nb] = buf_load(desc_bias, col_ok ? n * 4 : (int)OOB);
      if (n_res && p.res_rows == 0) {
        const int mlane = m0 + wm + 4 * half;
        const int vbase = col_ok ? (mlane * p.ldr + n) * 4 : (int)OOB;
        const int rows_left = p.M - mlane, ldr4 = p.ldr * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          rv[nb][r] = buf_load_s(desc_res, c < rows_left ? vbase : (int)OOB, c * ldr4);
        }
      } else if (n_res) {
        const int rr0 = m0 % p.res_rows;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dm = wm + mfma32_row(r, half);
          int rr = rr0 + dm;
          if (p.res_rows >= BM) rr = rr >= p.res_rows ? rr - p.res_rows : rr;
          else rr %= p.res_rows;
          rv[nb][r] = buf_load(desc_res, (col_ok && m0 + dm < p.M) ? (rr * p.ldr + n) * 4 : (int)OOB);
        }
      }
    }
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    const Item it = item_of(c_i);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[nb][r] += acc[2 + nb][r]; acc[2 + nb][r] = 0.f; }
    const int m0 = it.bm * BM;
    const int mlane = m0 + wm + 4 * half;
    const int rows_left = p.M - mlane;
    if (SK) {            // the raw partial tile -> the slice's slab [M][Cout]
      const i32x4 desc_slab = raw_desc(scratch + (long)(it.kt0 / nk) * p.M * p.Cout, (long)p.M * p.Cout * 4);
      const int lds4 = p.Cout * 4;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int n = it.bn * BN + wn + 32 * nb + l31;
        const int vbase_s = n < p.Cout ? (mlane * p.Cout + n) * 4 : (int)OOB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          buf_store_s(desc_slab, c < rows_left ? vbase_s : (int)OOB, c * lds4, acc[nb][r]);
          acc[nb][r] = 0.f;
        }
      }
      stores_pending = true;
      return;
    }
    // residual and bias were fetched at the head of this phase
    if (n_res | n_bias) __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));
    const int ldc4 = p.ldc * 4;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = it.bn * BN + wn + 32 * nb + l31;
      const bool col_ok = n < p.Cout;
      if (n_bias) asm volatile("" : "+v"(bv[nb]));
      if (n_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(rv[nb][r]));
      }
      const int vbase = col_ok ? (mlane * p.ldc + n) * 4 : (int)OOB;
      if (n_bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] += bv[nb];
      }
      if (n_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] += rv[nb][r];
      }
      with_act(p.act, [&](auto ACT) __attribute__((always_inline)) -> void {
        constexpr int act = decltype(ACT)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          buf_store_s(desc_out, c < rows_left ? vbase : (int)OOB, c * ldc4, apply_act(acc[nb][r], act));
          acc[nb][r] = 0.f;
        }
      });
    }
    stores_pending = true;
  };

  auto phase_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);                     // nothing -- MFMAs and splits included -- moves across a phase boundary
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // LOAD(s): fragments of ring stage SLOT -> registers, A rows split; the DMA of step s+2 -> stage SLOT + 2
  auto load_phase = [&](auto SLOT) __attribute__((always_inline)) -> void {
    constexpr int slot = decltype(SLOT)::value, islot = (slot + 2) % 3;
    x6w_fetch_a(ra, aaddr, (unsigned)(slot * STAGE_BYTES));
    x6w_fetch_b<B_PIECE>(rb, baddr + (unsigned)(slot * STAGE_BYTES));
    issue(std::integral_constant<int, islot>{});
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 10));       // LDS reads return in order: the four A chunks are the oldest
    x6w_landed_a(ra);
    split3(ra[0], ra[1], ap[0]);
    split3(ra[2], ra[3], ap[1]);
    // the planes exist HERE, in this phase (hipcc otherwise sinks the split across the barrier to the MFMAs that use it)
#pragma unroll
    for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(ap[s][0]), "+v"(ap[s][1]), "+v"(ap[s][2]));
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
    x6w_landed_b(rb);
    if (grp) {           // Y: its pieces of step s+1 have landed before X reads them in the next phase
      __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW, 15));
      stores_pending = false;
    }
  };
  // COMPUTE(s): 24 MFMAs, the four accumulators in turn (consecutive MFMAs are independent); smallest terms first per accumulator
  auto compute_phase = [&](auto SLOT) __attribute__((always_inline)) -> void {
    constexpr int islot = (decltype(SLOT)::value + 2) % 3;
    if (c_kt == nk - 1 && (n_res | n_bias)) epi_loads();   // last k-step of the tile: its residual and bias fly under the MFMAs
#define AOT_PP_TERM(PA, PB)                                                                                        \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                \
      acc[2 * s + nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[s][PA], rb[PB][s][nb], acc[2 * s + nb], 0, 0, 0);
    AOT_PP_TERM(1, 1)
    AOT_PP_TERM(0, 2)
    AOT_PP_TERM(2, 0)
    AOT_PP_TERM(0, 1)
    AOT_PP_TERM(1, 0)
    AOT_PP_TERM(0, 0)
#undef AOT_PP_TERM
    bool did_epi = false;
    if (++c_kt == nk) {
      epilogue();
      c_kt = 0;
      ++c_i;
      did_epi = true;
    }
    if (!grp) {          // X: its pieces of step s+1 have landed before anyone reads them in the next phase
      if (did_epi) __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW + 32, 15));
      else __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW, 15));
      stores_pending = false;
    }
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  issue(I0{});
  issue(I1{});
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW, 15));          // step 0 has landed
  phase_barrier();
  if (grp) phase_barrier();                                  // Y starts one phase late
  auto step = [&](auto U) __attribute__((always_inline)) -> void {
    load_phase(U);
    phase_barrier();
    compute_phase(U);
    phase_barrier();
  };
#pragma unroll 1
  for (int ss = 0; ss < total; ss += 3) {
    step(I0{});
    if (ss + 1 < total) step(I1{});
    if (ss + 2 < total) step(I2{});
  }
  if (!grp) phase_barrier();                                 // X waits for Y's last phase (same barrier count in both groups)
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));             // the all-out-of-bounds DMAs past the end still target this LDS
}

}  // namespace

// the bf16 x 6 kernel takes what the lean kernel takes (32-bit operand offsets) with K a multiple of 32
bool gemm_x6_eligible(const ConvParams& p) {
  return (p.Cin % 32) == 0 && (p.K % 32) == 0 && (p.lda & 3) == 0 && ((uintptr_t)p.in & 15) == 0 &&
         (long)p.B * p.H * p.W * p.lda * 4 < 0x7fffffffL && (long)p.M * p.ldc * 4 < 0x7fffffffL &&
         (!p.res || (long)(p.res_rows ? p.res_rows : p.M) * p.ldr * 4 < 0x7fffffffL);
}

// the phase-shifted 128x128 form with split-K over the grid (gemm_x6pp_kernel<., true>): slabs [ksplit][M][Cout] in `scratch`
// (its unsplit form was no faster than the plain 128x128 kernel -- profiles/r05_x6pp.txt -- and is not built)
int launch_gemm_x6pp(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s, int ksplit, float* scratch, const LnOutArgs* ln) {
  if (!gemm_x6_eligible(p) || !w6 || (cout_pad % 64) || cout_pad < p.Cout || ((uintptr_t)w6 & 15)) return AOT_ERR_UNSUPPORTED;
  if (3L * (p.K / 8) * cout_pad * 16 >= 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  if (ksplit < 2 || (p.K / BK) % ksplit != 0) return AOT_ERR_BADARG;
  if (!scratch || (long)p.M * p.Cout * 4 >= 0x7fffffffL) return AOT_ERR_BADARG;
  const bool is1x1 = (p.KH == 1 && p.KW == 1 && p.pad == 0);
  X6Weight wq;
  wq.w6 = w6;
  wq.cout_pad = cout_pad;
  const int nit = cdiv(p.M, 128) * cdiv(p.Cout, 128) * ksplit;
  const int grid = nit < 256 ? nit : 256;                     // one 8-wave workgroup per CU
  if (is1x1)
    hipLaunchKernelGGL((gemm_x6pp_kernel<true, true>), dim3(grid), dim3(512), 0, s, p, wq, ksplit, scratch);
  else
    hipLaunchKernelGGL((gemm_x6pp_kernel<false, true>), dim3(grid), dim3(512), 0, s, p, wq, ksplit, scratch);
  if (ln) launch_splitk_reduce_ln(p, ksplit, scratch, ln->gamma, ln->beta, ln->out, ln->ld, ln->eps, s);
  else launch_splitk_reduce(p, ksplit, scratch, s);
  AOT_LAUNCH_CHECK();
}

// linear layer on the 64x64 direct-weight kernel whose tile end also writes GroupNorm partial sums (gemm_x6rd_kernel<true, false, true>)
int launch_gemm_x6rd_gn(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s, float* gn_part) {
  if (!gemm_x6_eligible(p) || !w6 || (cout_pad % 64) || cout_pad < p.Cout || ((uintptr_t)w6 & 15)) return AOT_ERR_UNSUPPORTED;
  if (3L * (p.K / 8) * cout_pad * 16 >= 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  if (!gn_part || (p.Cout & 31) || !(p.KH == 1 && p.KW == 1 && p.pad == 0)) return AOT_ERR_BADARG;
  X6Weight wq;
  wq.w6 = w6;
  wq.cout_pad = cout_pad;
  const int nit = cdiv(p.M, 64) * cdiv(p.Cout, 64);
  const int grid = nit < 768 ? nit : 768;
  hipLaunchKernelGGL((gemm_x6rd_kernel<true, false, true>), dim3(grid), dim3(256), 0, s, p, wq, 1, gn_part, 0.f, nullptr, X6Group{});
  AOT_LAUNCH_CHECK();
}

// LayerNorm + linear layer in one launch (gemm_x6rd_kernel<true, false, GN, false, true>): `in` is the un-normalised [M, K] map, w6 the
// planes of W' = diag(gamma) W, bias = beta W + b, colsum = the column sums of W' (all folded by the caller); gn_part != nullptr: the
// GroupNorm partials of the output as well
int launch_gemm_x6rd_ln(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s, float eps, const float* colsum, float* gn_part) {
  if (!gemm_x6_eligible(p) || !w6 || (cout_pad % 64) || cout_pad < p.Cout || ((uintptr_t)w6 & 15)) return AOT_ERR_UNSUPPORTED;
  if (3L * (p.K / 8) * cout_pad * 16 >= 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  if (!(p.KH == 1 && p.KW == 1 && p.pad == 0 && p.stride == 1) || !(eps > 0.f) || !colsum) return AOT_ERR_BADARG;
  if (gn_part && (p.Cout & 31)) return AOT_ERR_BADARG;
  X6Weight wq;
  wq.w6 = w6;
  wq.cout_pad = cout_pad;
  const int nit = cdiv(p.M, 64) * cdiv(p.Cout, 64);
  const int grid = nit < 768 ? nit : 768;
  if (gn_part)
    hipLaunchKernelGGL((gemm_x6rd_kernel<true, false, true, false, true>), dim3(grid), dim3(256), 0, s, p, wq, 1, gn_part, eps, colsum, X6Group{});
  else
    hipLaunchKernelGGL((gemm_x6rd_kernel<true, false, false, false, true>), dim3(grid), dim3(256), 0, s, p, wq, 1, nullptr, eps, colsum, X6Group{});
  AOT_LAUNCH_CHECK();
}

// n <= 4 linear layers of one shape in one launch (gemm_x6rd_kernel<true, false, false, false, false, true>, blockIdx.y = the problem)
int launch_gemm_x6rd_group(const ConvParams& p, int n, const float* const* in, const void* const* w6, const float* const* bias,
                           const float* const* res, float* const* out, int cout_pad, hipStream_t s) {
  if (n < 1 || n > 4 || !in || !w6 || !out || (cout_pad % 64) || cout_pad < p.Cout) return AOT_ERR_BADARG;
  if (!(p.KH == 1 && p.KW == 1 && p.pad == 0 && p.stride == 1)) return AOT_ERR_BADARG;
  if (3L * (p.K / 8) * cout_pad * 16 >= 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  X6Group grp{};
  ConvParams q = p;
  for (int g = 0; g < n; ++g) {
    q.in = in[g]; q.out = out[g]; q.res = res ? res[g] : nullptr;
    if (!in[g] || !w6[g] || !out[g] || ((uintptr_t)w6[g] & 15) || !gemm_x6_eligible(q)) return AOT_ERR_UNSUPPORTED;
    if ((res && res[g] != nullptr) != (res && res[0] != nullptr) || (bias && bias[g] != nullptr) != (bias && bias[0] != nullptr))
      return AOT_ERR_BADARG;      // bias / residual: for all problems or for none (the kernel tests the pointers of problem 0)
    grp.in[g] = in[g]; grp.w6[g] = w6[g]; grp.bias[g] = bias ? bias[g] : nullptr; grp.res[g] = res ? res[g] : nullptr; grp.out[g] = out[g];
  }
  q.in = in[0]; q.out = out[0]; q.res = res ? res[0] : nullptr; q.bias = bias ? bias[0] : nullptr;
  X6Weight wq;
  wq.w6 = w6[0];
  wq.cout_pad = cout_pad;
  const int nit = cdiv(p.M, 64) * cdiv(p.Cout, 64);
  const int gx = nit < 768 ? nit : 768;
  hipLaunchKernelGGL((gemm_x6rd_kernel<true, false, false, false, false, true>), dim3(gx, n), dim3(256), 0, s, q, wq, 1, nullptr, 0.f, nullptr, grp);
  AOT_LAUNCH_CHECK();
}

// a KxK convolution on FOUR input channels (the ResNet stem) on the 64x64 direct-weight kernel: w6 = the planes of the weight
// [ceil(KH * KW / 8) * 32, ld] (rows k = 4 * tap + channel, zero rows past KH * KW * 4)
int launch_gemm_x6rd_c4(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s) {
  if (!w6 || (cout_pad % 64) || cout_pad < p.Cout || ((uintptr_t)w6 & 15) || ((uintptr_t)p.in & 15)) return AOT_ERR_UNSUPPORTED;
  if (p.Cin != 4 || p.lda != 4 || p.KH * p.KW <= 1) return AOT_ERR_UNSUPPORTED;
  const int nk = (p.KH * p.KW + 7) / 8;
  if ((long)p.B * p.H * p.W * 16 >= 0x7fffffffL || (long)p.M * p.ldc * 4 >= 0x7fffffffL || 3L * nk * 4 * cout_pad * 16 >= 0x7fffffffL ||
      (p.res && (long)(p.res_rows ? p.res_rows : p.M) * p.ldr * 4 >= 0x7fffffffL))
    return AOT_ERR_UNSUPPORTED;
  X6Weight wq;
  wq.w6 = w6;
  wq.cout_pad = cout_pad;
  const int nit = cdiv(p.M, 64) * cdiv(p.Cout, 64);
  const int grid = nit < 768 ? nit : 768;
  hipLaunchKernelGGL((gemm_x6rd_kernel<false, false, false, true>), dim3(grid), dim3(256), 0, s, p, wq, 1, nullptr, 0.f, nullptr, X6Group{});
  AOT_LAUNCH_CHECK();
}

// split-K over the grid on the 64x64 register-staged kernel with direct weight fragments (gemm_x6rd_kernel<., true>)
int launch_gemm_x6rd_splitk(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s, int ksplit, float* scratch, const LnOutArgs* ln) {
  if (!gemm_x6_eligible(p) || !w6 || (cout_pad % 64) || cout_pad < p.Cout || ((uintptr_t)w6 & 15)) return AOT_ERR_UNSUPPORTED;
  if (3L * (p.K / 8) * cout_pad * 16 >= 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  if (ksplit < 2 || (p.K / BK) % ksplit != 0 || !scratch || (long)p.M * p.Cout * 4 >= 0x7fffffffL) return AOT_ERR_BADARG;
  X6Weight wq;
  wq.w6 = w6;
  wq.cout_pad = cout_pad;
  const int nit = cdiv(p.M, 64) * cdiv(p.Cout, 64) * ksplit;
  const int grid = nit < 1024 ? nit : 1024;                  // (the split-K form needs 118 registers: four workgroups per CU)
  if (p.KH == 1 && p.KW == 1 && p.pad == 0)
    hipLaunchKernelGGL((gemm_x6rd_kernel<true, true>), dim3(grid), dim3(256), 0, s, p, wq, ksplit, scratch, 0.f, nullptr, X6Group{});
  else
    hipLaunchKernelGGL((gemm_x6rd_kernel<false, true>), dim3(grid), dim3(256), 0, s, p, wq, ksplit, scratch, 0.f, nullptr, X6Group{});
  if (ln) launch_splitk_reduce_ln(p, ksplit, scratch, ln->gamma, ln->beta, ln->out, ln->ld, ln->eps, s);
  else launch_splitk_reduce(p, ksplit, scratch, s);
  AOT_LAUNCH_CHECK();
}

// The bf16x6 product (terms = 6) by shape, or one member by name (`tile`; tests and tools/dev/mb_gemm.py): 66 = the 64x64 register-staged
// kernel with the weight fragments straight from global memory (the family's default), 129 = the register-staged 128x128 tile (the
// KxK layers that fill exactly one dispatch round with it).  terms = 1: the
// plain bf16 product of the training path (gemm_bf16_kernel), optionally split-K.
int launch_gemm_x6(const ConvParams& p, const void* w6, int cout_pad, int tile, hipStream_t s, int terms, int ksplit, float* scratch) {
  if (!gemm_x6_eligible(p) || !w6 || (cout_pad % 64) || cout_pad < p.Cout || ((uintptr_t)w6 & 15)) return AOT_ERR_UNSUPPORTED;
  if (3L * (p.K / 8) * cout_pad * 16 >= 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  const bool is1x1 = (p.KH == 1 && p.KW == 1 && p.pad == 0);
  X6Weight wq;
  wq.w6 = w6;
  wq.cout_pad = cout_pad;
  if (terms == 1) {           // plain bf16 (training): the 64x64 LDS-DMA tile, two workgroups per CU
    if (ksplit < 1) return AOT_ERR_BADARG;
    const int nit = cdiv(p.M, 64) * cdiv(p.Cout, 64);
    const int g1 = nit < 512 ? nit : 512;
    if (ksplit > 1) {         // split-K (weight gradients): 1x1 only, K / 32 divisible, partial slabs + the reduce pass
      if (!is1x1 || (p.K / BK) % ksplit != 0 || !scratch || (long)p.M * p.Cout * 4 >= 0x7fffffffL) return AOT_ERR_BADARG;
      const int gk = nit * ksplit < 512 ? nit * ksplit : 512;
      hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), dim3(gk), dim3(256), 0, s, p, wq, ksplit, scratch);
      launch_splitk_reduce(p, ksplit, scratch, s);
      AOT_LAUNCH_CHECK();
    }
    if (is1x1)
      hipLaunchKernelGGL((gemm_bf16_kernel<true>), dim3(g1), dim3(256), 0, s, p, wq, 1, nullptr);
    else
      hipLaunchKernelGGL((gemm_bf16_kernel<false>), dim3(g1), dim3(256), 0, s, p, wq, 1, nullptr);
    AOT_LAUNCH_CHECK();
  }
  if (terms != 6) return AOT_ERR_BADARG;
  if (ksplit != 1) return AOT_ERR_BADARG;
  if (tile == 0) {
    // Round 5 (profiles/r05_x6r.txt, r05_x6r128.txt, r05_x6rd.txt: every conv / linear of the frame at batch 1 and 3): the 64x64
    // direct-weight kernel is the default of the family; the 128x128 tile keeps the KxK layers that fill exactly ONE dispatch round
    // with it (>= 200 tiles: the 3x3 convolutions of the decoder at the 4x map), whose activation rows it re-reads half as often
    // across the filter taps
    const int nwide = cdiv(p.M, 128) * cdiv(p.Cout, 128);
    tile = (p.KH * p.KW > 1 && p.Cout >= 128 && nwide >= 200 && nwide <= 256) ? 129 : 66;
  }
  if (tile == 66) {             // the register-staged 64x64 form with the weight fragments straight from global memory
    const int nit = cdiv(p.M, 64) * cdiv(p.Cout, 64);
    const int gr = nit < 768 ? nit : 768;
    if (is1x1)
      hipLaunchKernelGGL((gemm_x6rd_kernel<true, false>), dim3(gr), dim3(256), 0, s, p, wq, 1, nullptr, 0.f, nullptr, X6Group{});
    else
      hipLaunchKernelGGL((gemm_x6rd_kernel<false, false>), dim3(gr), dim3(256), 0, s, p, wq, 1, nullptr, 0.f, nullptr, X6Group{});
    AOT_LAUNCH_CHECK();
  }
  if (tile == 129) {            // the register-staged 128x128 form: eight waves, one workgroup per CU
    const int nit = cdiv(p.M, 128) * cdiv(p.Cout, 128);
    const int gr = nit < 256 ? nit : 256;
    if (is1x1)
      hipLaunchKernelGGL((gemm_x6r_kernel<true, 4, 2>), dim3(gr), dim3(512), 0, s, p, wq);
    else
      hipLaunchKernelGGL((gemm_x6r_kernel<false, 4, 2>), dim3(gr), dim3(512), 0, s, p, wq);
    AOT_LAUNCH_CHECK();
  }
  return AOT_ERR_BADARG;
}
