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
#include "conv_params.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

namespace {

constexpr int BK = 32, PF = 2, NSTAGE = PF + 1;
constexpr int GROUP_STRIDE = 8 * 128 + 16;   // bytes: 8 rows x 32 floats + one 16-byte pad

__device__ __forceinline__ int chunk_off(int row, int c) {
  const int g = row >> 3, r = row & 7;
  return g * GROUP_STRIDE + r * 128 + ((c ^ r) << 4);
}

template <int LPW>   // LDS-DMA instructions per wave and k-step
__device__ __forceinline__ void wait_steps_in_flight(int n) {
  // s_waitcnt simm16 on gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14 ; here expcnt/lgkmcnt = no wait
  if (LPW == 6) {
    switch (n) {
      case 0: __builtin_amdgcn_s_waitcnt(0xF70); break;
      case 1: __builtin_amdgcn_s_waitcnt(0xF76); break;
      default: __builtin_amdgcn_s_waitcnt(0xF7C); break;
    }
  } else {   // 4
    switch (n) {
      case 0: __builtin_amdgcn_s_waitcnt(0xF70); break;
      case 1: __builtin_amdgcn_s_waitcnt(0xF74); break;
      default: __builtin_amdgcn_s_waitcnt(0xF78); break;
    }
  }
}

// Fragment reads are plain LDS loads (the compiler places their lgkmcnt waits).  hipcc drains every LDS-DMA in flight
// (vmcnt(0)) in front of a plain LDS load, so inside a k-step all fragment reads of step s+1 are issued BEFORE the DMA of
// step s+2: at that point nothing is in flight (the counted vmcnt wait + barrier have just made step s+1 visible).
__device__ __forceinline__ float4 lds_read128(const unsigned char* base, int off) {
  return *reinterpret_cast<const float4*>(base + off);
}

struct Item {
  int bm, bn, kt0;   // tile coordinates and first k-step of the slice
};

template <int BMB, bool IS1X1>
__global__ void __launch_bounds__(256, BMB == 2 ? 2 : 3) gemm_lds_kernel(const ConvParams p, const int ksplit, float* __restrict__ scratch) {
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
  float4 ra[2][BMB][4], rb[2][4];     // [register set][block][chunk]
  auto fetch = [&](int set, int slot) {
    const unsigned char* st = lds + slot * STAGE_BYTES;
#pragma unroll
    for (int x = 0; x < BMB; ++x)
#pragma unroll
      for (int j = 0; j < 4; ++j) ra[set][x][j] = lds_read128(st, aoff[x][j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) rb[set][j] = lds_read128(st, boff[j]);
  };

  f32x16 acc[BMB];
#pragma unroll
  for (int x = 0; x < BMB; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

#pragma unroll 1
  for (int i = 0; i < PF && issued < total; ++i) issue_next();
  wait_steps_in_flight<LPW>(min(PF - 1, total - 1));
  __builtin_amdgcn_s_barrier();
  fetch(0, 0);

  int c_i = 0, c_kt = 0, rd_slot = 1;      // rd_slot: ring slot of step s+1
  auto mfma_half = [&](int set, int j0) {
#pragma unroll
    for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
      for (int x = 0; x < BMB; ++x) {
        const float4 a4 = ra[set][x][j], b4 = rb[set][j];
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
        if (ss + 1 < total) {
          wait_steps_in_flight<LPW>(0);             // step ss+1 has landed (the only DMA in flight)
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

// sum of the k-slices in slice order + epilogue; one thread per 4 output channels
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ConvParams p, const int ksplit, const float* __restrict__ scratch) {
  const int nq = (p.Cout + 3) >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.M * nq) return;
  const int m = (int)(idx / nq), n0 = (int)(idx - (long)m * nq) * 4;
  const long slab = (long)p.M * p.Cout;
  const float* src = scratch + (long)m * p.Cout + n0;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const int cnt = min(4, p.Cout - n0);
  for (int s = 0; s < ksplit; ++s)
    for (int c = 0; c < cnt; ++c) v[c] += src[(long)s * slab + c];
  const long rrow = p.res_rows ? m % p.res_rows : m;
  for (int c = 0; c < cnt; ++c) {
    float t = v[c] + (p.bias ? p.bias[n0 + c] : 0.f);
    if (p.res) t += p.res[rrow * p.ldr + n0 + c];
    p.out[(long)m * p.ldc + n0 + c] = apply_act(t, p.act);
  }
}

template <int BMB>
int launch_variant(const ConvParams& p, bool is1x1, int ksplit, float* scratch, hipStream_t s) {
  constexpr int BM = 64 * BMB;
  const int nitems = cdiv(p.M, BM) * cdiv(p.Cout, 64) * ksplit;
  const int per_cu = BMB == 2 ? 2 : 3;       // resident workgroups per CU (LDS: 73 KB / 49 KB per workgroup)
  const int grid = nitems < 256 * per_cu ? nitems : 256 * per_cu;
  if (is1x1)
    hipLaunchKernelGGL((gemm_lds_kernel<BMB, true>), dim3(grid), dim3(256), 0, s, p, ksplit, scratch);
  else
    hipLaunchKernelGGL((gemm_lds_kernel<BMB, false>), dim3(grid), dim3(256), 0, s, p, ksplit, scratch);
  if (ksplit > 1) {
    const long n = (long)p.M * ((p.Cout + 3) >> 2);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, ksplit, scratch);
  }
  AOT_LAUNCH_CHECK();
}

}  // namespace

bool gemm_lds_eligible(const ConvParams& p) {
  return p.wt != nullptr && (p.Cin % 32) == 0 && (p.K % 32) == 0 && (p.lda & 3) == 0 && (p.ldwt & 3) == 0 &&
         ((uintptr_t)p.wt & 15) == 0;
}

int launch_gemm_lds(const ConvParams& p, int variant, int ksplit, float* scratch, hipStream_t s) {
  if (!gemm_lds_eligible(p)) return AOT_ERR_UNSUPPORTED;
  if (ksplit < 1 || (p.K / BK) % ksplit != 0) return AOT_ERR_BADARG;
  if (ksplit > 1 && !scratch) return AOT_ERR_BADARG;
  const bool is1x1 = (p.KH == 1 && p.KW == 1 && p.pad == 0);
  if (variant == 0) return launch_variant<2>(p, is1x1, ksplit, scratch, s);
  if (variant == 1) return launch_variant<1>(p, is1x1, ksplit, scratch, s);
  return AOT_ERR_BADARG;
}
