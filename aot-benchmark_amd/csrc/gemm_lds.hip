// LDS-direct implicit-GEMM convolution / linear layer on the exact-fp32 matrix cores of gfx950 (the main GEMM path:
// every conv / linear whose Cin is a multiple of 32; gemm_conv.hip keeps the register-staged kernel for the stem, the
// MobileNet shapes and Cout <= 32).
//
//   out[m, n] = act( sum_k A[m, k] * Wt[n, k] + bias[n] + res[m % res_rows, n] )
//
// A is the im2col view of B NHWC images (m = (b, oy, ox), k = (ky, kx, c)), never materialised; Wt is the weight with
// k-contiguous rows ([Cout, KH*KW*Cin], FrozenBN folded in by the host).  A 256-thread workgroup owns a BM x 64 tile
// (BM = 128: four waves of 64x32 = two 32x32 MFMA blocks sharing their B fragment; BM = 64: four waves of 32x32) and
// walks K in steps of 32 through a 3-stage LDS ring that is filled by global_load_lds_dwordx4 -- no staging VGPRs, no
// ds_write pass:
//   * LDS image of an operand: groups of 8 rows x 128 bytes (+16 bytes of padding per group); one wave-level LDS-DMA
//     fills one group (64 lanes x 16 bytes, lane -> row lane>>3, 16-byte slot lane&7).  The DMA writes lane-linear, so
//     the XOR swizzle that makes the fragment reads conflict-free is applied to the SOURCE address (slot s of row r
//     receives k-chunk s^r) and again on the read (cdna_hip_programming.md rule 21).
//   * fragments are read with ds_read_b128 (four consecutive k of one row); the k order inside a step is therefore
//     permuted, identically for A and B.
//   * pipeline per k-step s:  wait until step s+1 has landed (vmcnt) ; s_barrier ; read the fragments of step s+1 into
//     the second register set ; issue the DMA of step s+2 ; 32 (16) MFMAs of step s.  The DMA of step s+2 flies under
//     the MFMAs of steps s and, for a second resident workgroup, under that workgroup's whole step.
//     The issue side runs ahead across tile boundaries (persistent workgroups, XCD-aware item order).
//   * im2col: a 32-wide k-step never straddles a filter tap (Cin % 32 == 0), so the tap is wave-uniform per step and a
//     lane only decides whether its row's tap lies inside the image; out-of-image taps and rows >= M read a zero page.
//   * split-K (ksplit > 1): item = (tile, k-slice); partial tiles go to an fp32 slab and splitk_reduce_kernel sums the
//     slices in slice order (deterministic) and applies the epilogue.  Used when a shape has too few tiles for 256 CUs.
// Numerics: v_mfma_f32_32x32x2_f32 is an exact k-ordered fp32 fmaf chain; results differ from an fp32 reference by
// summation order only.
#include "gemm_tile.h"

__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

bool gemm_lean_eligible(const ConvParams& p);

namespace {

template <int LPW, bool LGKM0>   // LPW = LDS-DMA instructions per wave and k-step; LGKM0: also wait for every LDS read
__device__ __forceinline__ void wait_steps_in_flight(int n) {
  constexpr int L = LGKM0 ? 0 : 15;
  switch (n) {
    case 0: __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, L)); break;
    case 1: __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW, L)); break;
    case 2: __builtin_amdgcn_s_waitcnt(waitcnt_imm(2 * LPW, L)); break;
    default: __builtin_amdgcn_s_waitcnt(waitcnt_imm(3 * LPW, L)); break;
  }
}

// Fragment reads are plain LDS loads (the compiler places their lgkmcnt waits).  hipcc drains every LDS-DMA in flight
// (vmcnt(0)) in front of a plain LDS load, so inside a k-step all fragment reads of step s+1 are issued BEFORE the DMA of
// step s+2: at that point nothing is in flight (the counted vmcnt wait + barrier have just made step s+1 visible).
__device__ __forceinline__ f32x4 lds_read128(const unsigned char* base, int off) {
  return *reinterpret_cast<const f32x4*>(base + off);
}


// PFD = k-steps of LDS-DMA kept ahead of the MFMAs.  PFD = 2 is the pipeline described above (plain LDS loads; one step
// of MFMA time to cover the DMA latency, hidden by the 2-3 workgroups that share a CU).  PFD = 3 / 4 is for shapes with
// about one workgroup per CU or fewer, where nothing else covers that latency: the fragment reads become inline-asm
// ds_read_b128 (a plain LDS load makes hipcc drain every DMA in flight), ordered by hand -- a counted vmcnt +
// lgkmcnt(0) wait in front of the barrier of each step -- so 2 / 3 steps of DMA stay in flight across the barrier.
template <int BMB, bool IS1X1, int PFD>
__global__ void __launch_bounds__(256, (BMB == 2 ? (PFD > 2 ? 1 : 2) : (PFD > 3 ? 1 : PFD > 2 ? 2 : 3)))
gemm_lds_kernel(const ConvParams p, const int ksplit, float* __restrict__ scratch) {
  constexpr int PF = PFD, NSTAGE = PFD + 1;
  constexpr bool ASM_READS = PFD > 2;
  constexpr int BM = 64 * BMB, BN = 64;
  constexpr int AG = BM / 8, BG = BN / 8;             // 8-row groups per operand tile
  constexpr int AGW = AG / 4, BGW = BG / 4;           // groups filled by each wave
  constexpr int LPW = AGW + BGW;
  constexpr int OPA_BYTES = AG * GROUP_STRIDE;
  constexpr int STAGE_BYTES = (AG + BG) * GROUP_STRIDE;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSTAGE * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int nbn = (p.Cout + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
  const int nk_all = p.K / BK, nk = nk_all / ksplit;   // k-steps per item (host guarantees divisibility)
  const int nitems = nbm * nbn * ksplit;
  // XCD-aware item order: workgroup w runs on XCD w % 8; every XCD owns one contiguous run of items (items of one row
  // panel and k-slice are adjacent and share their A rows through that XCD's L2), walked round-robin by its workgroups.
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
  const int nwg_x = ((int)gridDim.x - xcd + 7) >> 3;
  const int q8 = nitems >> 3, r8 = nitems & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
  const int mine = li < chunk_n ? (chunk_n - li + nwg_x - 1) / nwg_x : 0;
  const int total = mine * nk;
  if (total == 0) return;
  auto item_of = [&](int i) {   // i-th item of this workgroup; item = ((bm * ksplit) + ks) * nbn + bn
    const int it = chunk0 + li + i * nwg_x;
    Item r;
    r.bn = it % nbn;
    const int t = it / nbn;
    r.kt0 = (t % ksplit) * nk;
    r.bm = t / ksplit;
    return r;
  };

  const int wm = (wave >> 1) * 32 * BMB, wn = (wave & 1) * 32;
  const int lr = lane >> 3, lp = lane & 7;
  const int cofs = (lp ^ lr) << 2;        // k offset (floats) of the 16-byte chunk this lane fetches
  const float* zero = g_zero_page;
  const int hw_out = p.OH * p.OW;

  // ---- issue side --------------------------------------------------------------------------------------------------
  int is_i = 0, is_kt = 0, is_slot = 0, issued = 0;
  const float* a_ptr[AGW];     // 1x1: row base + cofs (nullptr-like zero page if the row is >= M)
  int a_iy0[AGW], a_ix0[AGW];  // KxK: top-left input coordinate of the row's window
  bool a_ok[AGW];
  const float* b_ptr[BGW];
  int tap_c = 0, tap_ky = 0, tap_kx = 0;
  auto setup_item = [&](int i) {
    const Item it = item_of(i);
#pragma unroll
    for (int g = 0; g < AGW; ++g) {
      const int m = it.bm * BM + 8 * (AGW * wave + g) + lr;
      a_ok[g] = m < p.M;
      const int mm = a_ok[g] ? m : 0;
      const int b = mm / hw_out, pix = mm - b * hw_out;
      const int oy = pix / p.OW, ox = pix - oy * p.OW;
      const float* img = p.in + (long)b * p.H * p.W * p.lda;
      if (IS1X1) {
        a_ptr[g] = a_ok[g] ? img + ((long)(oy * p.stride) * p.W + ox * p.stride) * p.lda + it.kt0 * BK + cofs : zero;
      } else {
        a_ptr[g] = img + cofs;
        a_iy0[g] = oy * p.stride - p.pad;
        a_ix0[g] = ox * p.stride - p.pad;
      }
    }
#pragma unroll
    for (int g = 0; g < BGW; ++g) {
      const int n = it.bn * BN + 8 * (BGW * wave + g) + lr;
      b_ptr[g] = n < p.Cout ? p.wt + (long)n * p.ldwt + it.kt0 * BK + cofs : zero;
    }
    if (!IS1X1) {
      const int k0 = it.kt0 * BK;
      const int tap = k0 / p.Cin;
      tap_c = k0 - tap * p.Cin;
      tap_ky = tap / p.KW;
      tap_kx = tap - tap_ky * p.KW;
    }
  };
  auto issue_next = [&]() {
    if (is_kt == 0) setup_item(is_i);
    unsigned char* sa = lds + is_slot * STAGE_BYTES;
    unsigned char* sb = sa + OPA_BYTES;
#pragma unroll
    for (int g = 0; g < AGW; ++g) {
      const float* src;
      if (IS1X1) {
        src = a_ptr[g];
        if (a_ok[g]) a_ptr[g] += BK;
      } else {
        const int iy = a_iy0[g] + tap_ky * p.dil, ix = a_ix0[g] + tap_kx * p.dil;
        const bool in = a_ok[g] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        src = in ? a_ptr[g] + ((long)iy * p.W + ix) * p.lda + tap_c : zero;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + (AGW * wave + g) * GROUP_STRIDE), 16, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < BGW; ++g) {
      __builtin_amdgcn_global_load_lds((gptr_t)b_ptr[g], (lptr_t)(sb + (BGW * wave + g) * GROUP_STRIDE), 16, 0, 0);
      if (b_ptr[g] != zero) b_ptr[g] += BK;
    }
    if (!IS1X1) {
      tap_c += BK;
      if (tap_c >= p.Cin) { tap_c = 0; if (++tap_kx == p.KW) { tap_kx = 0; ++tap_ky; } }
    }
    ++issued;
    if (++is_kt == nk) { is_kt = 0; ++is_i; }
    if (++is_slot == NSTAGE) is_slot = 0;
  };

  // ---- compute side ------------------------------------------------------------------------------------------------
  int aoff[BMB][4], boff[4];     // A: [32-row block of the wave tile][chunk j]; B: one 32-column block
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int x = 0; x < BMB; ++x) aoff[x][j] = chunk_off(wm + 32 * x + l31, 2 * j + half);
    boff[j] = OPA_BYTES + chunk_off(wn + l31, 2 * j + half);
  }
  f32x4 ra[2][BMB][4], rb[2][4];      // [register set][block][chunk]
  const unsigned lds_base = (unsigned)(size_t)(lptr_t)lds;
  auto fetch = [&](int set, int slot) {
    if (ASM_READS) {        // asynchronous: the registers are valid after the next lgkmcnt(0) wait (see `landed`)
      const unsigned st = lds_base + (unsigned)(slot * STAGE_BYTES);
#pragma unroll
      for (int x = 0; x < BMB; ++x)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(ra[set][x][j]) : "v"(st + (unsigned)aoff[x][j]));
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(rb[set][j]) : "v"(st + (unsigned)boff[j]));
      return;
    }
    const unsigned char* st = lds + slot * STAGE_BYTES;
#pragma unroll
    for (int x = 0; x < BMB; ++x)
#pragma unroll
      for (int j = 0; j < 4; ++j) ra[set][x][j] = lds_read128(st, aoff[x][j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) rb[set][j] = lds_read128(st, boff[j]);
  };
  // makes the fragment registers of `set` depend on the wait that has just retired their ds_reads
  auto landed = [&](int set) {
#pragma unroll
    for (int x = 0; x < BMB; ++x)
      asm volatile("" : "+v"(ra[set][x][0]), "+v"(ra[set][x][1]), "+v"(ra[set][x][2]), "+v"(ra[set][x][3]));
    asm volatile("" : "+v"(rb[set][0]), "+v"(rb[set][1]), "+v"(rb[set][2]), "+v"(rb[set][3]));
  };

  f32x16 acc[BMB];
#pragma unroll
  for (int x = 0; x < BMB; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

#pragma unroll 1
  for (int i = 0; i < PF && issued < total; ++i) issue_next();
  wait_steps_in_flight<LPW, false>(min(PF - 1, total - 1));
  __builtin_amdgcn_s_barrier();
  fetch(0, 0);

  int c_i = 0, c_kt = 0, rd_slot = 1;      // rd_slot: ring slot of step s+1
  auto mfma_half = [&](int set, int j0) {
#pragma unroll
    for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
      for (int x = 0; x < BMB; ++x) {
        const f32x4 a4 = ra[set][x][j], b4 = rb[set][j];
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[x], 0, 0, 0);
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[x], 0, 0, 0);
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[x], 0, 0, 0);
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[x], 0, 0, 0);
      }
  };
  auto epilogue = [&]() {
    const Item it = item_of(c_i);
    const int n = it.bn * BN + wn + l31;
    const bool col_ok = n < p.Cout;
    if (ksplit > 1) {
      float* dst = scratch + (long)(it.kt0 / nk) * p.M * p.Cout;
#pragma unroll
      for (int x = 0; x < BMB; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = it.bm * BM + wm + 32 * x + mfma32_row(r, half);
          if (col_ok && m < p.M) dst[(long)m * p.Cout + n] = acc[x][r];
          acc[x][r] = 0.f;
        }
      return;
    }
    const float bv = (col_ok && p.bias) ? p.bias[n] : 0.f;
    const int m0 = it.bm * BM;
    const int rr0 = p.res_rows ? m0 % p.res_rows : m0;          // residual row of the tile's first row (one modulo per tile)
    const int wrap = p.res_rows ? p.res_rows : 0x7fffffff;
#pragma unroll
    for (int x = 0; x < BMB; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dm = wm + 32 * x + mfma32_row(r, half);
        const int m = m0 + dm;
        if (col_ok && m < p.M) {
          float v = acc[x][r] + bv;
          if (p.res) {
            int rr = rr0 + dm;
            while (rr >= wrap) rr -= wrap;
            v += p.res[(long)rr * p.ldr + n];
          }
          p.out[(long)m * p.ldc + n] = apply_act(v, p.act);
        }
        acc[x][r] = 0.f;
      }
  };

#pragma unroll 1
  for (int s = 0; s < total; s += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {                 // unrolled by two so the register-set index is static
      const int ss = s + u;
      if (ss < total) {
        if (ASM_READS) {
          if (ss + 1 < total) {
            // steps ss+1 .. min(ss+PF-1, total-1) are in flight: wait for step ss+1 (and for this wave's reads of step ss)
            wait_steps_in_flight<LPW, true>(min(PF - 2, total - 2 - ss));
            __builtin_amdgcn_s_barrier();           // step ss+1 visible to every wave; the slot of step ss-1 is free
            landed(u);
            fetch(u ^ 1, rd_slot);                  // fragments of step ss+1 -> the other register set (asynchronous)
            if (++rd_slot == NSTAGE) rd_slot = 0;
            if (issued < total) issue_next();       // DMA of step ss+PF into the slot of step ss-1
          } else {
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
            landed(u);
          }
        } else if (ss + 1 < total) {
          wait_steps_in_flight<LPW, false>(0);      // step ss+1 has landed (the only DMA in flight)
          __builtin_amdgcn_s_barrier();             // ... for every wave; every wave is done reading the slot of step ss-1
          fetch(u ^ 1, rd_slot);                    // fragments of step ss+1 -> the other register set
          if (++rd_slot == NSTAGE) rd_slot = 0;
          if (issued < total) issue_next();         // DMA of step ss+2 into the slot of step ss-1
        }
        mfma_half(u, 0);
        mfma_half(u, 2);
        if (++c_kt == nk) {
          epilogue();
          c_kt = 0;
          ++c_i;
        }
      }
    }
  }
}


// asynchronous fragment reads of one ring stage (IMM = byte offset of the stage relative to the base registers): the
// registers are valid after the next lgkmcnt(0) wait; frags_landed ties them to that wait for the compiler
template <int BMB, int IMM>
__device__ __forceinline__ void fetch_frags(f32x4 (&a)[BMB][4], f32x4 (&b)[4], const unsigned (&aaddr)[BMB][4],
                                            const unsigned (&baddr)[4]) {
#pragma unroll
  for (int x = 0; x < BMB; ++x)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[x][j]) : "v"(aaddr[x][j]), "n"(IMM));
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[j]) : "v"(baddr[j]), "n"(IMM));
}
template <int IMM>
__device__ __forceinline__ void fetch_one(f32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}
template <int BMB>
__device__ __forceinline__ void frags_landed(f32x4 (&a)[BMB][4], f32x4 (&b)[4]) {
#pragma unroll
  for (int x = 0; x < BMB; ++x) asm volatile("" : "+v"(a[x][0]), "+v"(a[x][1]), "+v"(a[x][2]), "+v"(a[x][3]));
  asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}

// ---- lean variant ---------------------------------------------------------------------------------------------------
// Same tile, LDS image and numerics class as gemm_lds_kernel, rebuilt around what the counters say about it
// (profiles/r02_gemm_pmc.txt): with one workgroup per CU a k-step spent 1024 cycles in its 16 MFMAs, ~760 cycles ISSUING the
// other ~140 instructions of the step (64-bit im2col address arithmetic, the zero-page selects, ring-slot bookkeeping, a
// register-set copy the compiler re-rolled the loop into) and ~640 cycles waiting for a DMA it had issued one step before.
//   * operands come through BUFFER loads to LDS (buffer_load_dwordx4 ... lds): a lane's address is a 32-bit byte offset
//     (+ a wave-uniform SGPR offset that walks K), rows >= M, columns >= Cout and filter taps outside the image get the
//     offset 0x80000000, which the descriptor's bounds check turns into zeros -- no pointers, no zero page, no selects on
//     64-bit values; a 1x1 layer spends no VALU instruction per step on addresses, a KxK layer six per 8-row group;
//   * four ring stages, three steps of DMA ahead; steps past the end of the workgroup's work are issued all-out-of-bounds, so
//     every step issues the same number of DMAs and the counted waits are compile-time constants;
//   * the loop is unrolled by four with the stage and the register set static: fragment reads are inline-asm ds_read_b128
//     with immediate stage offsets (a plain LDS load would make hipcc drain the DMA queue), retired by one
//     `s_waitcnt vmcnt(LPW) lgkmcnt(0)` in front of each step's barrier;
//   * the tile end has ONE wave-uniform branch on the activation per tile (apply_act()'s if-chain per output element -- five
//     scalar compare-and-branch pairs for each of 16-32 elements per lane -- was as long as the 32 MFMAs of a K = 64 tile), the
//     row part of every store / residual address is a scalar offset (one integer multiply per tile), bias and residual are
//     added in passes of their own: measured +2.5 % on the whole frame (round 3, profiles/r03a_variants.txt);
//   * the 64x64 tile keeps TWO accumulators per wave (even / odd half of each k-step, summed in the epilogue): consecutive
//     MFMAs are independent, so the step's few remaining instructions can sit between them (an instruction between two
//     MFMAs on ONE accumulator costs ~43 cycles on gfx950, between independent ones ~6).
template <int BMB, bool IS1X1>
__global__ void __launch_bounds__(256, (BMB == 2 ? 1 : 2))
gemm_lean_kernel(const ConvParams p, const int ksplit, float* __restrict__ scratch) {
  constexpr int NST = 4;
  constexpr int BM = 64 * BMB, BN = 64;
  constexpr int AG = BM / 8, BG = BN / 8, AGW = AG / 4, BGW = BG / 4, LPW = AGW + BGW;
  constexpr int OPA_BYTES = AG * GROUP_STRIDE, STAGE_BYTES = (AG + BG) * GROUP_STRIDE;
  constexpr int NACC = BMB == 1 ? 2 : BMB;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int nbn = (p.Cout + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
  const int nk = (p.K / BK) / ksplit;
  const int nitems = nbm * nbn * ksplit;
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
    const int t = it / nbn;
    r.kt0 = (t % ksplit) * nk;
    r.bm = t / ksplit;
    return r;
  };
  const int wm = (wave >> 1) * 32 * BMB, wn = (wave & 1) * 32;
  const int lr = lane >> 3, lp = lane & 7;
  const int cofs = (lp ^ lr) << 2;
  const int hw_out = p.OH * p.OW;
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)((long)p.B * p.H * p.W * p.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, (int)((long)p.Cout * p.ldwt * 4), 0x00020000);

  // epilogue operands through descriptors too (offset 0x80000000 = masked: loads give 0, stores are dropped), so that
  // the number of vector-memory instructions a tile end issues is FIXED and the counted waits below stay exact
  const i32x4 desc_out = raw_desc(p.out, (long)p.M * p.ldc * 4);
  const i32x4 desc_res = raw_desc(p.res, (long)(p.res_rows ? p.res_rows : p.M) * p.ldr * 4);
  const i32x4 desc_bias = raw_desc(p.bias, (long)p.Cout * 4);
  constexpr bool PREFETCH_EPI = BMB == 1;      // residual + bias of a tile are fetched under its last k-step (64x64 tile only:
                                               // the 128x64 tile has no registers to spare for them)
  const bool fused_epi = ksplit == 1;
  const int n_res = (fused_epi && p.res) ? 16 * BMB : 0, n_bias = (fused_epi && p.bias) ? 1 : 0;

  // ---- issue side --------------------------------------------------------------------------------------------------
  int is_i = 0, is_kt = 0;
  int a_off[AGW];              // byte offset of the row's window origin (+ this lane's 16-byte chunk); may be < 0 with padding
  int a_iy0[AGW], a_ix0[AGW];
  bool a_ok[AGW];
  unsigned b_off[BGW];
  int s_k = 0;                 // wave-uniform byte offset along K (B rows, and A rows of a 1x1 layer)
  int tap_c = 0, tap_ky = 0, tap_kx = 0;
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
#pragma unroll
    for (int g = 0; g < BGW; ++g) {
      const int n = it.bn * BN + 8 * (BGW * wave + g) + lr;
      b_off[g] = (live && n < p.Cout) ? (unsigned)((n * p.ldwt + cofs) * 4) : OOB;
    }
    s_k = it.kt0 * BK * 4;
    if (!IS1X1) {
      const int k0 = it.kt0 * BK;
      const int tap = k0 / p.Cin;
      tap_c = k0 - tap * p.Cin;
      tap_ky = tap / p.KW;
      tap_kx = tap - tap_ky * p.KW;
    }
  };
  // one step's DMA in pieces, so that the compute side can place them between its MFMAs
  int s_tap = 0;
  auto issue_begin = [&]() __attribute__((always_inline)) {
    if (is_kt == 0) setup_item(is_i);
    if (!IS1X1) s_tap = ((tap_ky * p.dil * p.W + tap_kx * p.dil) * p.lda + tap_c) * 4;
  };
  auto issue_a = [&](auto SLOT, auto G) __attribute__((always_inline)) -> void {
    constexpr int slot = decltype(SLOT)::value, g = decltype(G)::value;
    unsigned char* dst = lds + slot * STAGE_BYTES + (AGW * wave + g) * GROUP_STRIDE;
    if (IS1X1) {
      dma16(rsrc_a, dst, a_off[g], s_k);
    } else {
      const int iy = a_iy0[g] + tap_ky * p.dil, ix = a_ix0[g] + tap_kx * p.dil;
      const bool in = a_ok[g] & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      dma16(rsrc_a, dst, in ? a_off[g] + s_tap : (int)OOB, 0);
    }
  };
  auto issue_b = [&](auto SLOT, auto G) __attribute__((always_inline)) -> void {
    constexpr int slot = decltype(SLOT)::value, g = decltype(G)::value;
    dma16(rsrc_b, lds + slot * STAGE_BYTES + OPA_BYTES + (BGW * wave + g) * GROUP_STRIDE, (int)b_off[g], s_k);
  };
  auto issue_end = [&]() __attribute__((always_inline)) {
    s_k += BK * 4;
    if (!IS1X1) {
      tap_c += BK;
      if (tap_c >= p.Cin) { tap_c = 0; if (++tap_kx == p.KW) { tap_kx = 0; ++tap_ky; } }
    }
    if (++is_kt == nk) { is_kt = 0; ++is_i; }
  };
  auto issue = [&](auto SLOT) __attribute__((always_inline)) -> void {        // whole step at once (prologue)
    issue_begin();
    issue_a(SLOT, std::integral_constant<int, 0>{});
    issue_a(SLOT, std::integral_constant<int, 1>{});
    if constexpr (AGW > 2) {
      issue_a(SLOT, std::integral_constant<int, 2>{});
      issue_a(SLOT, std::integral_constant<int, 3>{});
    }
    issue_b(SLOT, std::integral_constant<int, 0>{});
    issue_b(SLOT, std::integral_constant<int, 1>{});
    issue_end();
  };

  // ---- compute side ------------------------------------------------------------------------------------------------
  const unsigned lds_base = (unsigned)(size_t)(lptr_t)lds;
  unsigned aaddr[2][BMB][4], baddr[2][4];     // [stages 0-1 / stages 2-3][block][chunk]: ds_read base registers
#pragma unroll
  for (int hs = 0; hs < 2; ++hs)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int x = 0; x < BMB; ++x)
        aaddr[hs][x][j] = lds_base + 2 * hs * STAGE_BYTES + chunk_off(wm + 32 * x + l31, 2 * j + half);
      baddr[hs][j] = lds_base + 2 * hs * STAGE_BYTES + OPA_BYTES + chunk_off(wn + l31, 2 * j + half);
    }
  f32x4 ra[2][BMB][4], rb[2][4];
  auto fetch = [&](auto SET, auto SLOT) __attribute__((always_inline)) -> void {
    constexpr int set = decltype(SET)::value, slot = decltype(SLOT)::value;
    fetch_frags<BMB, (slot & 1) * STAGE_BYTES>(ra[set], rb[set], aaddr[slot >> 1], baddr[slot >> 1]);
  };
  auto landed = [&](auto SET) __attribute__((always_inline)) -> void { frags_landed<BMB>(ra[decltype(SET)::value], rb[decltype(SET)::value]); };
  f32x16 acc[NACC];
#pragma unroll
  for (int x = 0; x < NACC; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  int c_i = 0, c_kt = 0;
  int stores_pending = 0;      // vector stores the previous step's epilogue issued (they count on vmcnt like the DMAs)
  float rv[BMB][16], bv = 0.f;
  // residual rows: row (m % res_rows) of a map shared by the lanes -- one modulo per tile, then a conditional subtract per
  // element (maps smaller than a tile take the general modulo)
  auto epi_loads = [&]() __attribute__((always_inline)) {
    const Item it = item_of(c_i);
    const int n = it.bn * BN + wn + l31;
    const bool col_ok = n < p.Cout;
    const int m0 = it.bm * BM;
    if (n_bias) bv = buf_load(desc_bias, col_ok ? n * 4 : (int)OOB);
    if (n_res && p.res_rows == 0) {          // a residual row per output row: lane offset once, row offsets as scalars
      const int mlane = m0 + wm + 4 * half;
      const int vbase = col_ok ? (mlane * p.ldr + n) * 4 : (int)OOB;
      const int rows_left = p.M - mlane, ldr4 = p.ldr * 4;
#pragma unroll
      for (int x = 0; x < BMB; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = 32 * x + (r & 3) + 8 * (r >> 2);
          rv[x][r] = buf_load_s(desc_res, c < rows_left ? vbase : (int)OOB, c * ldr4);
        }
    } else
    if (n_res) {
      const int rr0 = p.res_rows ? m0 % p.res_rows : m0;
#pragma unroll
      for (int x = 0; x < BMB; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dm = wm + 32 * x + mfma32_row(r, half);
          int rr = rr0 + dm;
          if (p.res_rows) {
            if (p.res_rows >= BM) rr = rr >= p.res_rows ? rr - p.res_rows : rr;
            else rr %= p.res_rows;
          }
          rv[x][r] = buf_load(desc_res, (col_ok && m0 + dm < p.M) ? (rr * p.ldr + n) * 4 : (int)OOB);
        }
    }
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    const Item it = item_of(c_i);
    const int n = it.bn * BN + wn + l31;
    const bool col_ok = n < p.Cout;
    if (BMB == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] += acc[1][r]; acc[1][r] = 0.f; }
    }
    if (!fused_epi) {        // split-K: the partial tile goes to its slab (plain stores: the next step waits for all of them)
      float* dst = scratch + (long)(it.kt0 / nk) * p.M * p.Cout;
#pragma unroll
      for (int x = 0; x < BMB; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = it.bm * BM + wm + 32 * x + mfma32_row(r, half);
          if (col_ok && m < p.M) dst[(long)m * p.Cout + n] = acc[x][r];
          acc[x][r] = 0.f;
        }
      return;
    }
    if (!PREFETCH_EPI) epi_loads();
    // the tile's residual and bias have arrived: with the prefetch they are older than the DMA pieces of step ss+3 (which
    // may stay in flight), without it they are the youngest instructions
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(PREFETCH_EPI ? LPW : 0, 15));
    if (n_bias) asm volatile("" : "+v"(bv));
    if (n_res) {
#pragma unroll
      for (int x = 0; x < BMB; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(rv[x][r]));
    }
    const int m0 = it.bm * BM;
    {
      const int mlane = m0 + wm + 4 * half;          // output row of accumulator register 0 (block 0) in this lane
      const int vbase = col_ok ? (mlane * p.ldc + n) * 4 : (int)OOB;
      const int rows_left = p.M - mlane, ldc4 = p.ldc * 4;
      // bias and residual in passes of their own (a wave-uniform branch each; same order of additions as the fused form)
      if (n_bias) {
#pragma unroll
        for (int x = 0; x < BMB; ++x)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[x][r] += bv;
      }
      if (n_res) {
#pragma unroll
        for (int x = 0; x < BMB; ++x)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[x][r] += rv[x][r];
      }
      auto store_all = [&](auto ACT) __attribute__((always_inline)) -> void {
        constexpr int act = decltype(ACT)::value;
#pragma unroll
        for (int x = 0; x < BMB; ++x)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = 32 * x + (r & 3) + 8 * (r >> 2);
            buf_store_s(desc_out, c < rows_left ? vbase : (int)OOB, c * ldc4, apply_act(acc[x][r], act));
            acc[x][r] = 0.f;
          }
      };
      switch (p.act) {
        case AOT_ACT_RELU: store_all(std::integral_constant<int, AOT_ACT_RELU>{}); break;
        case AOT_ACT_RELU6: store_all(std::integral_constant<int, AOT_ACT_RELU6>{}); break;
        case AOT_ACT_GELU: store_all(std::integral_constant<int, AOT_ACT_GELU>{}); break;
        case AOT_ACT_SILU: store_all(std::integral_constant<int, AOT_ACT_SILU>{}); break;
        default: store_all(std::integral_constant<int, AOT_ACT_NONE>{}); break;
      }
    }
    stores_pending = 16 * BMB;
  };
  constexpr int NM = 16 * BMB;              // MFMAs of one step
  constexpr int NR = 4 * BMB + 4;           // fragment reads of one step
  // MFMA number I of a step.  64x64 tile: chunks 0,1 -> accumulator 0, chunks 2,3 -> accumulator 1, the two chains alternate;
  // 128x64 tile: the two 32-row blocks alternate.  Consecutive MFMAs are independent either way.
  auto mfma_one = [&](auto SET, auto I) __attribute__((always_inline)) -> void {
    constexpr int set = decltype(SET)::value, i = decltype(I)::value;
    if constexpr (BMB == 1) {
      constexpr int e = i >> 2, a = i & 1, j = 2 * a + ((i >> 1) & 1);
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[set][0][j][e], rb[set][j][e], acc[a], 0, 0, 0);
    } else {
      constexpr int e = i >> 3, x = i & 1, j = (i >> 1) & 3;
      acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[set][x][j][e], rb[set][j][e], acc[x], 0, 0, 0);
    }
  };
  // what goes into the shadow of MFMA number I (each MFMA keeps the matrix pipe busy for 64 cycles; an LDS-DMA piece costs
  // about that much issue time, a ds_read_b128 a fraction of it): first the NR fragment reads of step ss+1, one per MFMA,
  // then the DMA of step ss+3, one piece per MFMA
  auto filler = [&](auto U, auto I) __attribute__((always_inline)) -> void {
    constexpr int u = decltype(U)::value, i = decltype(I)::value;
    constexpr int nset = (u + 1) & 1, nslot = (u + 1) & 3, islot = (u + 3) & 3;
    constexpr int imm = (nslot & 1) * STAGE_BYTES, hs = nslot >> 1;
    if constexpr (i < 4 * BMB) {
      fetch_one<imm>(ra[nset][i >> 2][i & 3], aaddr[hs][i >> 2][i & 3]);
    } else if constexpr (i < NR) {
      fetch_one<imm>(rb[nset][(i - 4 * BMB) & 3], baddr[hs][(i - 4 * BMB) & 3]);
    } else if constexpr (i == NR) {
      issue_begin();
    } else if constexpr (i <= NR + AGW) {
      issue_a(std::integral_constant<int, islot>{}, std::integral_constant<int, i - NR - 1>{});
    } else if constexpr (i <= NR + AGW + BGW) {
      issue_b(std::integral_constant<int, islot>{}, std::integral_constant<int, i - NR - AGW - 1>{});
    } else if constexpr (i == NR + AGW + BGW + 1) {
      issue_end();
    }
  };
  static_assert(NR + AGW + BGW + 1 < NM, "the step's fillers must fit between its MFMAs");
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  issue(I0{});
  issue(I1{});
  issue(I2{});
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(2 * LPW, 15));
  __builtin_amdgcn_s_barrier();
  fetch(I0{}, I0{});
  // step ss (ring stage U = ss % 4, register set U % 2): steps ss+1 and ss+2 are in flight on entry
  auto step = [&](auto U) __attribute__((always_inline)) -> void {
    constexpr int u = decltype(U)::value;
    // step ss+1 has landed (and this wave's fragment reads of step ss): everything but the youngest LPW vector-memory
    // instructions -- plus the stores of a tile the previous step finished, which are younger than the DMA waited for
    if (stores_pending) {
      __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW + 16 * BMB, 0));
      stores_pending = 0;
    } else {
      __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW, 0));
    }
    __builtin_amdgcn_s_barrier();                          // ... for every wave; the stage of step ss-1 is free
    landed(std::integral_constant<int, u & 1>{});
    if (PREFETCH_EPI && fused_epi && c_kt == nk - 1) epi_loads();   // last k-step of the tile: its residual and bias, now
    static_for<NM>([&](auto I) __attribute__((always_inline)) -> void {
      mfma_one(std::integral_constant<int, u & 1>{}, I);
      filler(U, I);
      __builtin_amdgcn_sched_barrier(0);                   // keep this placement: nothing moves across
    });
    if (++c_kt == nk) {
      epilogue();
      c_kt = 0;
      ++c_i;
    }
  };
#pragma unroll 1
  for (int ss = 0; ss < total; ss += 4) {
    step(I0{});
    if (ss + 1 < total) step(I1{});
    if (ss + 2 < total) step(I2{});
    if (ss + 3 < total) step(I3{});
  }
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));           // the all-out-of-bounds DMAs past the end still target this LDS
}

// sum of the k-slices in slice order + epilogue; one thread per 4 output channels (16-byte loads and stores where the rows allow it:
// Cout % 4 == 0 makes every slab row 16-byte aligned).
// LNO (round 6; Cout == 256: a row of the result is exactly one wave): the launch also writes LayerNorm(result) to a second map -- the
// wave holds the row, so the statistics are two butterflies; same arithmetic and order as layernorm_kernel<1> (bit-identical to a
// LayerNorm launch on the stored result).  The LSTT block's linear2 (+ residual) followed by the stack's output norm
// (transformer.py:124-135, 359-362 in the reference).
// (Tried in round 6 and removed: the GroupNorm statistics of the result out of this launch -- per-workgroup partials in double, a
//  device-scope ticket, the last workgroup adds them up.  Correct and deterministic, but 0.6 % SLOWER on the whole frame than the
//  256-workgroup statistics pass it replaced: profiles/r06_fusions_ab2.txt.)
struct LnOut {
  const float* gamma;
  const float* beta;
  float* out;
  int ld;
  float eps;
};
template <bool LNO>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ConvParams p, const int ksplit, const float* __restrict__ scratch, const LnOut ln) {
  const int nq = (p.Cout + 3) >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.M * nq) return;
  const int m = (int)(idx / nq), n0 = (int)(idx - (long)m * nq) * 4;
  const long slab = (long)p.M * p.Cout;
  const float* src = scratch + (long)m * p.Cout + n0;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const int cnt = min(4, p.Cout - n0);
  const bool vec = (p.Cout & 3) == 0 && ((uintptr_t)scratch & 15) == 0;
  if (vec) {
    for (int s = 0; s < ksplit; ++s) {
      const float4 t = *reinterpret_cast<const float4*>(src + (long)s * slab);
      v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
  } else {
    for (int s = 0; s < ksplit; ++s)
      for (int c = 0; c < cnt; ++c) v[c] += src[(long)s * slab + c];
  }
  const long rrow = p.res_rows ? m % p.res_rows : m;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < cnt; ++c) {
    float t = v[c] + (p.bias ? p.bias[n0 + c] : 0.f);
    if (p.res) t += p.res[rrow * p.ldr + n0 + c];
    o[c] = apply_act(t, p.act);
  }
  float* dst = p.out + (long)m * p.ldc + n0;
  if (vec && (p.ldc & 3) == 0 && ((uintptr_t)p.out & 15) == 0) {
    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    for (int c = 0; c < cnt; ++c) dst[c] = o[c];
  }
  if (LNO) {          // nq == 64: the 64 lanes of this wave hold row m (every lane of the wave is live or none)
    float sm = (o[0] + o[1]) + (o[2] + o[3]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sm += __shfl_xor(sm, off);
    const float mean = sm / (float)p.Cout;
    const float a = o[0] - mean, b = o[1] - mean, c = o[2] - mean, d = o[3] - mean;
    float sq = (a * a + b * b) + (c * c + d * d);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = 1.f / sqrtf(sq / (float)p.Cout + ln.eps);
    const float4 g4 = *reinterpret_cast<const float4*>(ln.gamma + n0), b4 = *reinterpret_cast<const float4*>(ln.beta + n0);
    float4 y;
    y.x = (o[0] - mean) * rstd * g4.x + b4.x;
    y.y = (o[1] - mean) * rstd * g4.y + b4.y;
    y.z = (o[2] - mean) * rstd * g4.z + b4.z;
    y.w = (o[3] - mean) * rstd * g4.w + b4.w;
    *reinterpret_cast<float4*>(ln.out + (long)m * ln.ld + n0) = y;
  }
}

template <int BMB, int PFD>
int launch_variant(const ConvParams& p, bool is1x1, int ksplit, float* scratch, hipStream_t s) {
  constexpr int BM = 64 * BMB;
  const int nitems = cdiv(p.M, BM) * cdiv(p.Cout, 64) * ksplit;
  // resident workgroups per CU (LDS: (PFD + 1) stages of 24.4 KB / 16.3 KB per workgroup, 160 KB per CU)
  constexpr int per_cu = BMB == 2 ? (PFD > 2 ? 1 : 2) : (PFD > 3 ? 1 : PFD > 2 ? 2 : 3);
  const int grid = nitems < 256 * per_cu ? nitems : 256 * per_cu;
  if (is1x1)
    hipLaunchKernelGGL((gemm_lds_kernel<BMB, true, PFD>), dim3(grid), dim3(256), 0, s, p, ksplit, scratch);
  else
    hipLaunchKernelGGL((gemm_lds_kernel<BMB, false, PFD>), dim3(grid), dim3(256), 0, s, p, ksplit, scratch);
  if (ksplit > 1) {
    const long n = (long)p.M * ((p.Cout + 3) >> 2);
    hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(cdiv(n, 256)), dim3(256), 0, s, p, ksplit, scratch, LnOut{});
  }
  AOT_LAUNCH_CHECK();
}

template <int BMB>
int launch_lean(const ConvParams& p, bool is1x1, int ksplit, float* scratch, hipStream_t s) {
  constexpr int BM = 64 * BMB;
  if (!gemm_lean_eligible(p)) return AOT_ERR_UNSUPPORTED;     // 32-bit byte offsets: both operand spans below 2 GB
  const int nitems = cdiv(p.M, BM) * cdiv(p.Cout, 64) * ksplit;
  constexpr int per_cu = BMB == 2 ? 1 : 2;
  const int grid = nitems < 256 * per_cu ? nitems : 256 * per_cu;
  if (is1x1)
    hipLaunchKernelGGL((gemm_lean_kernel<BMB, true>), dim3(grid), dim3(256), 0, s, p, ksplit, scratch);
  else
    hipLaunchKernelGGL((gemm_lean_kernel<BMB, false>), dim3(grid), dim3(256), 0, s, p, ksplit, scratch);
  if (ksplit > 1) {
    const long n = (long)p.M * ((p.Cout + 3) >> 2);
    hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(cdiv(n, 256)), dim3(256), 0, s, p, ksplit, scratch, LnOut{});
  }
  AOT_LAUNCH_CHECK();
}

}  // namespace

bool gemm_lds_eligible(const ConvParams& p) {
  return p.wt != nullptr && (p.Cin % 32) == 0 && (p.K % 32) == 0 && (p.lda & 3) == 0 && (p.ldwt & 3) == 0 &&
         ((uintptr_t)p.wt & 15) == 0;
}

// the lean kernel addresses both operands through buffer descriptors with 32-bit byte offsets
bool gemm_lean_eligible(const ConvParams& p) {
  return gemm_lds_eligible(p) && (long)p.B * p.H * p.W * p.lda * 4 < 0x7fffffffL && (long)p.Cout * p.ldwt * 4 < 0x7fffffffL &&
         (long)p.M * p.ldc * 4 < 0x7fffffffL && (!p.res || (long)(p.res_rows ? p.res_rows : p.M) * p.ldr * 4 < 0x7fffffffL) &&
         ((uintptr_t)p.in & 15) == 0;
}

void launch_splitk_reduce(const ConvParams& p, int ksplit, const float* scratch, hipStream_t s) {
  const long n = (long)p.M * ((p.Cout + 3) >> 2);
  hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(cdiv(n, 256)), dim3(256), 0, s, p, ksplit, scratch, LnOut{});
}

// ... or LayerNorm(result) as a second output (Cout == 256, 16-byte aligned rows everywhere: checked by the caller)
void launch_splitk_reduce_ln(const ConvParams& p, int ksplit, const float* scratch, const float* gamma, const float* beta, float* ln_out,
                             int ld_ln, float eps, hipStream_t s) {
  LnOut ln;
  ln.gamma = gamma; ln.beta = beta; ln.out = ln_out; ln.ld = ld_ln; ln.eps = eps;
  const long n = (long)p.M * (p.Cout >> 2);
  hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(cdiv(n, 256)), dim3(256), 0, s, p, ksplit, scratch, ln);
}

int launch_gemm_lds(const ConvParams& p, int variant, int ksplit, float* scratch, hipStream_t s) {
  if (!gemm_lds_eligible(p)) return AOT_ERR_UNSUPPORTED;
  if (ksplit < 1 || (p.K / BK) % ksplit != 0) return AOT_ERR_BADARG;
  if (ksplit > 1 && !scratch) return AOT_ERR_BADARG;
  const bool is1x1 = (p.KH == 1 && p.KW == 1 && p.pad == 0);
  switch (variant) {       // tile rows / 64, DMA steps kept ahead
    case 0: return launch_variant<2, 2>(p, is1x1, ksplit, scratch, s);
    case 1: return launch_variant<1, 2>(p, is1x1, ksplit, scratch, s);
    case 2: return launch_variant<1, 3>(p, is1x1, ksplit, scratch, s);
    case 3: return launch_variant<1, 4>(p, is1x1, ksplit, scratch, s);
    case 4: return launch_variant<2, 3>(p, is1x1, ksplit, scratch, s);
    case 5: return launch_variant<2, 4>(p, is1x1, ksplit, scratch, s);
    case 6: return launch_lean<1>(p, is1x1, ksplit, scratch, s);
    case 7: return launch_lean<2>(p, is1x1, ksplit, scratch, s);
    default: return AOT_ERR_BADARG;
  }
}
