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
#include <cstdlib>
#include <type_traits>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

bool gemm_lean_eligible(const ConvParams& p);

namespace {

constexpr int BK = 32;
constexpr int GROUP_STRIDE = 8 * 128 + 16;   // bytes: 8 rows x 32 floats + one 16-byte pad

__device__ __forceinline__ int chunk_off(int row, int c) {
  const int g = row >> 3, r = row & 7;
  return g * GROUP_STRIDE + r * 128 + ((c ^ r) << 4);
}

// s_waitcnt simm16 on gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14 (expcnt: no wait)
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

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

struct Item {
  int bm, bn, kt0;   // tile coordinates and first k-step of the slice
};

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
// one LDS-DMA piece: 64 lanes x 16 bytes, lane address = descriptor base + voff (+ the wave-uniform soff); an offset beyond the
// descriptor's range (0x80000000) delivers zeros.  (A __device__ function of its own: the target builtin inside a generic
// lambda silently keeps hipcc's HOST pass from emitting the kernel's launch stub.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)dst, 16, voff, soff, 0, 0);
}
// raw buffer descriptor (stride 0) as four scalars, for the inline-asm buffer loads / stores of the epilogue
__device__ __forceinline__ i32x4 raw_desc(const void* base, long bytes) {
  const unsigned long long a = (unsigned long long)(uintptr_t)base;
  i32x4 d;
  d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  d[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
  d[2] = __builtin_amdgcn_readfirstlane(base ? (int)(bytes < 0x7fffffffL ? bytes : 0x7fffffffL) : 0);
  d[3] = 0x00020000;
  return d;
}
// asynchronous (the compiler does not see them as memory operations: the callers place the vmcnt waits).  The s_nop covers
// the "VALU writes an SGPR -> vector-memory instruction reads it" hazard (5 wait states), which hipcc cannot insert for an
// instruction hidden in inline asm: under register pressure it restores the descriptor from spill lanes with v_readlane
// right in front of the asm (seen in the KxK kernel: the load then went out with a stale base address).
__device__ __forceinline__ float buf_load(const i32x4& desc, int voff) {
  float v;
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(desc));
  return v;
}
__device__ __forceinline__ void buf_store(const i32x4& desc, int voff, float v) {
  asm volatile("s_nop 4\n\tbuffer_store_dword %0, %1, %2, 0 offen" : : "v"(v), "v"(voff), "s"(desc) : "memory");
}
// the same with a wave-uniform byte offset in an SGPR next to the lane's offset (address = base + soff + voff; an out-of-range
// voff still masks the access whatever soff is)
__device__ __forceinline__ float buf_load_s(const i32x4& desc, int voff, int soff) {
  float v;
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(desc), "s"(soff));
  return v;
}
__device__ __forceinline__ void buf_store_s(const i32x4& desc, int voff, int soff, float v) {
  asm volatile("s_nop 4\n\tbuffer_store_dword %0, %1, %2, %3 offen" : : "v"(v), "v"(voff), "s"(desc), "s"(soff) : "memory");
}
// bits [31:16] of v as one 16-bit store (a truncated-bf16 plane element)
__device__ __forceinline__ void buf_store_hi16(const i32x4& desc, int voff, int soff, float v) {
  asm volatile("s_nop 4\n\tbuffer_store_short_d16_hi %0, %1, %2, %3 offen" : : "v"(v), "v"(voff), "s"(desc), "s"(soff) : "memory");
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
#ifdef AOT_LEAN_TIMING      // developer build (tools/dev/gemm_check): where does a step spend its cycles?
  unsigned long long tm_wait = 0, tm_bar = 0, tm_body = 0, tm_prev = 0, tm_steps = 0;
#endif
  auto step = [&](auto U) __attribute__((always_inline)) -> void {
    constexpr int u = decltype(U)::value;
#ifdef AOT_LEAN_TIMING
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
    // step ss+1 has landed (and this wave's fragment reads of step ss): everything but the youngest LPW vector-memory
    // instructions -- plus the stores of a tile the previous step finished, which are younger than the DMA waited for
    if (stores_pending) {
      __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW + 16 * BMB, 0));
      stores_pending = 0;
    } else {
      __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW, 0));
    }
#ifdef AOT_LEAN_TIMING
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
#endif
    __builtin_amdgcn_s_barrier();                          // ... for every wave; the stage of step ss-1 is free
#ifdef AOT_LEAN_TIMING
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
    tm_wait += t2 - t1; tm_bar += t3 - t2;
    if (tm_prev) tm_body += t1 - tm_prev;
    tm_prev = t3; ++tm_steps;
#endif
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
#ifdef AOT_LEAN_TIMING
  if (tid == 0 && ksplit == 1 && scratch) {
    unsigned long long* c = reinterpret_cast<unsigned long long*>(scratch);
    atomicAdd(c + 0, tm_wait); atomicAdd(c + 1, tm_bar); atomicAdd(c + 2, tm_body); atomicAdd(c + 3, tm_steps);
  }
#endif
}

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

// NT = 6: the fp32-equivalent six-term product (inference, mfma = 'bf16x6').  NT = 1: ONE product of operands rounded to bf16 --
// the training path's `precision = 'bf16'` (train_ops.matmul_precision): weight plane from aot_pack_bf16_f32 (round to nearest
// even), activations rounded in registers; 2 MFMAs per k-step instead of 12, one weight plane through the ring instead of three.
// SK (with NT = 1, 1x1 only): split-K for the weight gradients of the training path -- item = (tile, k-slice), the partial tile goes
// raw to its fp32 slab of `scratch` [ksplit][M][Cout] and splitk_reduce_kernel sums the slabs in order (+ bias / residual / act).
// PS (with NT = 6): the activations arrive ALREADY SPLIT -- p.in is three bf16 planes [3][B*H*W][lda] (element stride lda, k in
// natural order; aot_split3_bf16_f32 or a producer's tile end), the weight planes in natural k order too (aot_pack_bf16x6n_f32).
// The A tile of a k-step is 3 x 64 rows x 64 bytes: one DMA piece per wave and plane (lane = (row, 16-byte chunk), the chunk
// XOR-swizzled by the row so that the fragment reads are conflict-free), and the fragment goes from LDS straight into the MFMA: no
// vector instruction touches it (the split is 7.3 of the 10.7 VALU per MFMA of the fp32-activation form: profiles/r04_x6_gemm_pmc.txt).
// OP (with PS): the tile end writes the result AS THREE bf16 PLANES [3][M][ldp] (natural channel order: the input format of this very
// member) instead of fp32 -- the producer side of a conv -> conv chain.  The planes' base arrives in `scratch`, ldp in `ksplit` (the
// split-K arguments, unused here: the kernel's argument layout stays what the shipped members were validated with).  Three
// buffer_store_short_d16_hi per value: bits [31:16] of v, of v - hi(v) and of that remainder's remainder (exact).
template <bool IS1X1, int NT = 6, bool SK = false, bool PS = false, bool OP = false>
__global__ void __launch_bounds__(256, 2) gemm_x6_kernel(const ConvParams p, const X6Weight wq, const int ksplit, float* __restrict__ scratch) {
  static_assert(!SK || (IS1X1 && NT == 1), "split-K: the plain bf16 1x1 member only");
  static_assert(!PS || (NT == 6 && !SK), "pre-split activations: the six-term member only");
  static_assert(!OP || PS, "plane output: the pre-split member only");
  constexpr int NSTORE = OP ? 48 : 16;                  // stores of a tile end per lane
  constexpr int NST = 3;
  constexpr int BM = 64, BN = 64;
  constexpr int AG = BM / 8, AGW = PS ? 1 : AG / 4;     // A: 8-row fp32 groups, two per wave (PS: one 16-row piece per plane)
  constexpr int BPW = NT == 1 ? 1 : 3;                  // B: one 16-byte chunk column (cc = wave) of each plane per wave
  constexpr int LPW = (PS ? 3 : AGW) + BPW;
  constexpr int A_PLANE = 64 * 64;                      // (PS) bytes of one plane of the A tile: 64 rows x 32 bf16
  constexpr int AEL = PS ? 2 : 4;                       // bytes per activation element
  constexpr int OPA_BYTES = PS ? 3 * A_PLANE : AG * GROUP_STRIDE, B_PIECE = 64 * 16, OPB_BYTES = 12 * B_PIECE;
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
  const int a_plane_bytes = p.B * p.H * p.W * p.lda * 2;         // (PS) one activation plane
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, PS ? 3 * a_plane_bytes : (int)((long)p.B * p.H * p.W * p.lda * 4),
                                        0x00020000);
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
      // fp32: lane = (row lr of the wave's g-th 8-row group, 16-byte chunk lp); PS: lane = (row lane >> 2 of the wave's 16 rows,
      // LDS chunk lane & 3, which holds the GLOBAL chunk (lane & 3) ^ ((row >> 1) & 3) of the 32-channel block)
      const int rowl = PS ? 16 * wave + (lane >> 2) : 8 * (AGW * wave + g) + lr;
      const int m = it.bm * BM + rowl;
      a_ok[g] = live && m < p.M;
      const int mm = a_ok[g] ? m : 0;
      const int b = mm / hw_out, pix = mm - b * hw_out;
      const int oy = pix / p.OW, ox = pix - oy * p.OW;
      a_iy0[g] = oy * p.stride - p.pad;
      a_ix0[g] = ox * p.stride - p.pad;
      const int celem = PS ? 8 * ((lane & 3) ^ ((rowl >> 1) & 3)) : cofs;
      a_off[g] = (((b * p.H + a_iy0[g]) * p.W + a_ix0[g]) * p.lda + celem) * AEL;
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
    if (PS) {          // one piece per plane: the wave's 16 rows x 64 bytes, lane-linear in LDS
      int voff = a_off[0];
      if (!IS1X1) {
        const int iy = a_iy0[0] + tap_ky * p.dil, ix = a_ix0[0] + tap_kx * p.dil;
        const bool in = a_ok[0] & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        voff = in ? a_off[0] + s_tap : (int)OOB;
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dma16(rsrc_a, st + pl * A_PLANE + wave * 1024, voff, (IS1X1 ? s_k : 0) + pl * a_plane_bytes);
    } else {
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
  bf16x8 pa[2][3][2];          // (PS) [register set][plane][sub-step]: the A planes as they lie in LDS
  unsigned paddr[2];           // (PS) the lane's row; chunk 2 s + half at its swizzled place
#pragma unroll
  for (int sx = 0; sx < 2; ++sx) paddr[sx] = lds_base + (wm + l31) * 64 + (((2 * sx + half) ^ (((wm + l31) >> 1) & 3)) << 4);
  auto fetch = [&](auto SET, auto SLOT) __attribute__((always_inline)) -> void {
    if (PS) {
      constexpr int ST = decltype(SLOT)::value * STAGE_BYTES, q = decltype(SET)::value;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pa[q][0][0]) : "v"(paddr[0]), "n"(ST));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pa[q][0][1]) : "v"(paddr[1]), "n"(ST));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pa[q][1][0]) : "v"(paddr[0]), "n"(ST + A_PLANE));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pa[q][1][1]) : "v"(paddr[1]), "n"(ST + A_PLANE));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pa[q][2][0]) : "v"(paddr[0]), "n"(ST + 2 * A_PLANE));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pa[q][2][1]) : "v"(paddr[1]), "n"(ST + 2 * A_PLANE));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[q][0][0]) : "v"(baddr), "n"(ST + 0 * B_PIECE));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[q][0][1]) : "v"(baddr), "n"(ST + 2 * B_PIECE));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[q][1][0]) : "v"(baddr), "n"(ST + 4 * B_PIECE));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[q][1][1]) : "v"(baddr), "n"(ST + 6 * B_PIECE));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[q][2][0]) : "v"(baddr), "n"(ST + 8 * B_PIECE));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[q][2][1]) : "v"(baddr), "n"(ST + 10 * B_PIECE));
    } else {
      x6_fetch<decltype(SLOT)::value * STAGE_BYTES, B_PIECE, NT>(ra[decltype(SET)::value], rb[decltype(SET)::value], aaddr, baddr);
    }
  };
  auto landed = [&](auto SET) __attribute__((always_inline)) -> void {
    if (PS) {
      constexpr int q = decltype(SET)::value;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(pa[q][pl][0]), "+v"(pa[q][pl][1]), "+v"(rb[q][pl][0]), "+v"(rb[q][pl][1]));
    } else {
      x6_landed<NT>(ra[decltype(SET)::value], rb[decltype(SET)::value]);
    }
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
    if (OP) {
      const int ldp = ksplit;
      const int oplane = p.M * ldp * 2;                   // bytes of one output plane
      const i32x4 desc_pl = raw_desc(scratch, 3L * oplane);
      const int vbase_p = col_ok ? (mlane * ldp + n) * 2 : (int)OOB;
      const int ldp2 = ldp * 2;
      with_act(p.act, [&](auto ACT) __attribute__((always_inline)) -> void {
        constexpr int act = decltype(ACT)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const int vo = c < rows_left ? vbase_p : (int)OOB;
          const float v0 = apply_act(acc[0][r], act);
          const float v1 = v0 - __uint_as_float(__float_as_uint(v0) & 0xffff0000u);
          const float v2 = v1 - __uint_as_float(__float_as_uint(v1) & 0xffff0000u);
          buf_store_hi16(desc_pl, vo, c * ldp2, v0);
          buf_store_hi16(desc_pl, vo, c * ldp2 + oplane, v1);
          buf_store_hi16(desc_pl, vo, c * ldp2 + 2 * oplane, v2);
          acc[0][r] = 0.f;
        }
      });
      stores_pending = NSTORE;
      return;
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
      if (PS) {        // the planes as they came: same six products, same order
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[set][1][s], rb[set][1][s], acc[s], 0, 0, 0);
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[set][0][s], rb[set][2][s], acc[s], 0, 0, 0);
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[set][2][s], rb[set][0][s], acc[s], 0, 0, 0);
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[set][0][s], rb[set][1][s], acc[s], 0, 0, 0);
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[set][1][s], rb[set][0][s], acc[s], 0, 0, 0);
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[set][0][s], rb[set][0][s], acc[s], 0, 0, 0);
        continue;
      }
      if (NT == 1) {
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(round8(ra[set][2 * s], ra[set][2 * s + 1]), rb[set][0][s], acc[s], 0, 0, 0);
        continue;
      }
      bf16x8 ap[3];
      split3(ra[set][2 * s], ra[set][2 * s + 1], ap);
      // smallest terms first; the two sub-steps feed two independent accumulators
      acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], rb[set][1][s], acc[s], 0, 0, 0);
      acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], rb[set][2][s], acc[s], 0, 0, 0);
      acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], rb[set][0][s], acc[s], 0, 0, 0);
      acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], rb[set][1][s], acc[s], 0, 0, 0);
      acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], rb[set][0][s], acc[s], 0, 0, 0);
      acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], rb[set][0][s], acc[s], 0, 0, 0);
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

// wide tile: A fragments double-buffered, weight fragments of the current step (planes x sub-steps x column blocks)
// (three stages of 41 KB: the stage offset does not fit the 16-bit immediate, it is added to the address registers)
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

// 128x128 tile, eight waves (4 along M x 2 along N, each 32 rows x 64 columns): the activation split of a wave -- the VALU work
// of this family -- now feeds TWO column blocks, 24 MFMAs per k-step against the same 88 split instructions as the 64x64 tile's 12
// (there the split, not the matrix pipe, set the pace: 1.25x the fp32 lean kernel instead of the 2.67x of the MFMA count).  One
// workgroup per CU (123.6 KB of ring), still two waves per SIMD.  The weight fragments are read at the start of their own step
// (single register set: 48 registers instead of 96); the second wave of the SIMD covers their latency.
template <bool IS1X1>
__global__ void __launch_bounds__(512, 2) gemm_x6w_kernel(const ConvParams p, const X6Weight wq) {
  constexpr int NST = 3;
  constexpr int BM = 128, BN = 128;
  constexpr int AG = BM / 8, AGW = AG / 8;              // A: 8-row groups, two per wave (eight waves)
  constexpr int BPW = 3;                                // B: one 16-byte chunk column (cc = wave) of each plane per wave
  constexpr int LPW = AGW + BPW;
  constexpr int OPA_BYTES = AG * GROUP_STRIDE, B_PIECE = 64 * 16, OPB_BYTES = 24 * B_PIECE;      // piece (pl, cc, column half)
  constexpr int STAGE_BYTES = OPA_BYTES + OPB_BYTES;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE_BYTES];

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
  const int n_res = p.res ? 32 : 0, n_bias = p.bias ? 1 : 0;

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
    // this wave's DMA column: chunk cc = wave & 3 of k-block 0, column half wave >> 2 of the tile (columns past the padded
    // width -- a 128-wide tile on a weight padded to 64 -- are masked)
    b_off = (live && it.bn * BN + (wave >> 2) * 64 < wq.cout_pad)
                ? (unsigned)(((wave & 3) * wq.cout_pad + it.bn * BN + (wave >> 2) * 64 + lane) * 16) : OOB;
    s_k = 0;
    s_kb = 0;
    if (!IS1X1) { tap_c = 0; tap_ky = 0; tap_kx = 0; }
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
  // piece (pl, cc = 2 s + half, column half wn / 64) at ((pl * 4 + cc) * 2 + wn / 64) * B_PIECE; column block nb: + nb * 32 columns
  const unsigned baddr = lds_base + OPA_BYTES + (half * 2 + (wn >> 6)) * B_PIECE + l31 * 16;
  f32x4 ra[2][4];              // [register set][16-byte chunk j]: sub-step s contracts chunks 2 s and 2 s + 1
  bf16x8 rb[3][2][2];          // [plane][sub-step][column block]
  auto fetch_a = [&](auto SET, auto SLOT) __attribute__((always_inline)) -> void {
    x6w_fetch_a(ra[decltype(SET)::value], aaddr, (unsigned)(decltype(SLOT)::value * STAGE_BYTES));
  };
  auto fetch_b = [&](auto SLOT) __attribute__((always_inline)) -> void {
    x6w_fetch_b<B_PIECE>(rb, baddr + (unsigned)(decltype(SLOT)::value * STAGE_BYTES));
  };
  f32x16 acc[4];               // [2 * sub-step + column block]
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  int c_i = 0, c_kt = 0;
  int stores_pending = 0;
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
    // residual and bias were fetched under the tile's last k-step: older than the DMA pieces issued in that step
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW, 15));
    const int m0 = it.bm * BM;
    const int mlane = m0 + wm + 4 * half;
    const int rows_left = p.M - mlane, ldc4 = p.ldc * 4;
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
    stores_pending = 32;
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  issue(I0{});
  issue(I1{});
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW, 15));          // step 0 has landed
  __builtin_amdgcn_s_barrier();
  fetch_a(I0{}, I0{});
  // step ss (ring stage U % 3, A register set U % 2; the loop is unrolled by six): on entry the A fragments of step ss are being
  // read into set U % 2, the DMA of step ss+1 is in flight
  auto step = [&](auto U) __attribute__((always_inline)) -> void {
    constexpr int u = decltype(U)::value, set = u & 1, slot = u % 3, nslot = (u + 1) % 3, islot = (u + 2) % 3;
    // step ss+1 has landed, and this wave's fragment reads of step ss (plus the stores of a tile the previous step finished)
    if (stores_pending) {
      __builtin_amdgcn_s_waitcnt(waitcnt_imm(32, 0));
      stores_pending = 0;
    } else {
      __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
    }
    __builtin_amdgcn_s_barrier();                          // ... for every wave; the stage of step ss-1 (= of step ss+2) is free
    x6w_landed_a(ra[set]);
    fetch_b(std::integral_constant<int, slot>{});             // this step's weight fragments (asynchronous)
    if (c_kt == nk - 1) epi_loads();     // last k-step of the tile: its residual and bias, now -- BEFORE this step's DMA pieces, so
                                         // that the epilogue's counted wait (all but the youngest LPW) covers them
    issue(std::integral_constant<int, islot>{});                                             // DMA of step ss+2
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));           // the weight fragments have landed (nothing else is on lgkmcnt)
    x6w_landed_b(rb);
    fetch_a(std::integral_constant<int, set ^ 1>{}, std::integral_constant<int, nslot>{});     // A fragments of step ss+1, under the MFMAs
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ap[3];
      split3(ra[set][2 * s], ra[set][2 * s + 1], ap);
      // smallest terms first; sub-steps and column blocks feed four independent accumulators
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        f32x16& a = acc[2 * s + nb];
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], rb[1][s][nb], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], rb[2][s][nb], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], rb[0][s][nb], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], rb[1][s][nb], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], rb[0][s][nb], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], rb[0][s][nb], a, 0, 0, 0);
      }
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
// every wave owns a 32-row x 32-column block, i.e. 32 rows of ONE 32-channel group: (sum, sum of squares) of the block's valid
// elements (fp32, the stored values themselves) -> gn_part[(2 * tile row + wave row) * (Cout / 32) + column block][2].  The consumer
// (gn_act_dwconv5_kernel<true>) adds the partials of a group in index order in double: the statistics pass over the whole map and its
// launch are gone (linear1 -> GN -> GELU -> dw5x5 of the LSTT's feed-forward, transformer.py:355-362 / basic.py:15-35).
// C4 (the ResNet stem, 7x7 stride 2 on the image padded to FOUR channels): Cin = 4 makes one 16-byte chunk of the A row exactly one
// filter tap (r, g, b, 0 of one input pixel), so a k-step is eight taps instead of 32 channels of one tap: the thread's two chunks are
// two taps with a bounds check each; K = KH * KW * 4 rounded up to 32 (the weight rows past it are zero: aot_pack_bf16x6_f32 of the
// zero-padded matrix).  The last big layer that was still on the fp32 matrix cores in bf16x6 engines.
template <bool IS1X1, bool SK, bool GN = false, bool C4 = false>
__global__ void __launch_bounds__(256, 3) gemm_x6rd_kernel(const ConvParams p, const X6Weight wq, const int ksplit, float* __restrict__ scratch) {
  static_assert(!(SK && GN), "GroupNorm partials come from the unsplit form");
  static_assert(!C4 || (!IS1X1 && !SK && !GN), "the four-channel form: a KxK layer, unsplit");
  constexpr int WM = 2, NBW = 1;
  constexpr int NT = 128 * WM;                            // threads: WM x 2 waves
  constexpr int BM = 32 * WM, BN = 64 * NBW;
  constexpr int A_PLANE = BM * 64;                        // bytes: BM rows x four 16-byte chunks (32 bf16)
  constexpr int A_BYTES = 3 * A_PLANE;
  constexpr int BUF = A_BYTES;                            // 12 KB: the activation planes only
  constexpr unsigned OOB = 0x80000000u;
  static_assert(NT == 4 * BM, "one A fragment per thread and k-step");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

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
    split3(__builtin_bit_cast(f32x4, sa[0]), __builtin_bit_cast(f32x4, sa[1]), pl3);
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
      if (has_bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nb][r] += bv[nb];
      }
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nb][r] += rv[nb][r];
      }
      float ps = 0.f, pq = 0.f;
      with_act(p.act, [&](auto ACT) __attribute__((always_inline)) -> void {
        constexpr int act = decltype(ACT)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const float v = apply_act(acc[0][nb][r], act);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_out, c < rows_left ? vbase : (int)OOB, c * ldc4, 0);
          if (GN && c < rows_left && col_ok) { ps += v; pq += v * v; }
          acc[0][nb][r] = 0.f;
        }
      });
      if (GN) {          // the wave's 32 x 32 block = 32 rows of one group: fixed butterfly over the 64 lanes, lane 0 writes
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { ps += __shfl_xor(ps, off); pq += __shfl_xor(pq, off); }
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
#ifndef AOT_PP_DMAC
#define AOT_PP_DMAC 0        // development switch: the DMA pieces of step s+2 are issued in COMPUTE(s) instead of LOAD(s)
#endif
#ifndef AOT_PP_PRIO
#define AOT_PP_PRIO 0        // development switch: s_setprio 1 around the MFMA phase
#endif
#ifndef AOT_PP_PROBE
#define AOT_PP_PROBE 0       // timing probes (WRONG results): 1 no A DMA, 2 no B DMA, 4 no split, 8 no fragment reads, 16 no MFMAs
#endif
template <bool IS1X1, bool SK>
__global__ void __launch_bounds__(512, 2) gemm_x6pp_kernel(const ConvParams p, const X6Weight wq, const int ksplit, float* __restrict__ scratch) {
  constexpr bool DMAC = AOT_PP_DMAC != 0;
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
      if (AOT_PP_PROBE & 1) continue;
      if (IS1X1) {
        dma16(rsrc_a, dst, a_off[g], s_k);
      } else {
        const int iy = a_iy0[g] + tap_ky * p.dil, ix = a_ix0[g] + tap_kx * p.dil;
        const bool in = a_ok[g] & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        dma16(rsrc_a, dst, in ? a_off[g] + s_tap : (int)OOB, 0);
      }
    }
#pragma unroll
    for (int pl = 0; pl < ((AOT_PP_PROBE & 2) ? 0 : 3); ++pl)
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
    // residual and bias were fetched at the head of this phase (DMAC: older than the DMA pieces issued in it)
    if (n_res | n_bias) __builtin_amdgcn_s_waitcnt(waitcnt_imm(DMAC ? LPW : 0, 15));
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
  // LOAD(s): fragments of ring stage SLOT -> registers, A rows split; (not DMAC) the DMA of step s+2 -> stage SLOT + 2
  auto load_phase = [&](auto SLOT) __attribute__((always_inline)) -> void {
    constexpr int slot = decltype(SLOT)::value, islot = (slot + 2) % 3;
    if (!(AOT_PP_PROBE & 8)) {
      x6w_fetch_a(ra, aaddr, (unsigned)(slot * STAGE_BYTES));
      x6w_fetch_b<B_PIECE>(rb, baddr + (unsigned)(slot * STAGE_BYTES));
    }
    if (!DMAC) issue(std::integral_constant<int, islot>{});
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 10));       // LDS reads return in order: the four A chunks are the oldest
    x6w_landed_a(ra);
    if (AOT_PP_PROBE & 4) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) { ap[0][pl] = __builtin_bit_cast(bf16x8, ra[pl]); ap[1][pl] = __builtin_bit_cast(bf16x8, ra[3 - pl]); }
    } else {
      split3(ra[0], ra[1], ap[0]);
      split3(ra[2], ra[3], ap[1]);
    }
    // the planes exist HERE, in this phase (hipcc otherwise sinks the split across the barrier to the MFMAs that use it)
#pragma unroll
    for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(ap[s][0]), "+v"(ap[s][1]), "+v"(ap[s][2]));
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
    x6w_landed_b(rb);
    if (grp) {           // Y: its pieces of step s+1 have landed before X reads them in the next phase
      if (DMAC) {
        if (stores_pending) __builtin_amdgcn_s_waitcnt(waitcnt_imm(32, 15));
        else __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));
      } else {
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(LPW, 15));
      }
      stores_pending = false;
    }
  };
  // COMPUTE(s): 24 MFMAs, the four accumulators in turn (consecutive MFMAs are independent); smallest terms first per accumulator
  auto compute_phase = [&](auto SLOT) __attribute__((always_inline)) -> void {
    constexpr int islot = (decltype(SLOT)::value + 2) % 3;
    if (c_kt == nk - 1 && (n_res | n_bias)) epi_loads();   // last k-step of the tile: its residual and bias fly under the MFMAs
    if (AOT_PP_PRIO) __builtin_amdgcn_s_setprio(1);
#define AOT_PP_TERM(PA, PB)                                                                                        \
  if (!(AOT_PP_PROBE & 16)) _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                   \
      acc[2 * s + nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[s][PA], rb[PB][s][nb], acc[2 * s + nb], 0, 0, 0);
    AOT_PP_TERM(1, 1)
    AOT_PP_TERM(0, 2)
    AOT_PP_TERM(2, 0)
    if (DMAC) issue(std::integral_constant<int, islot>{});
    AOT_PP_TERM(0, 1)
    AOT_PP_TERM(1, 0)
    AOT_PP_TERM(0, 0)
#undef AOT_PP_TERM
    if (AOT_PP_PRIO) __builtin_amdgcn_s_setprio(0);
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

// sum of the k-slices in slice order + epilogue; one thread per 4 output channels (16-byte loads and stores where the rows allow it:
// Cout % 4 == 0 makes every slab row 16-byte aligned)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ConvParams p, const int ksplit, const float* __restrict__ scratch) {
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
  float o[4];
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
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, ksplit, scratch);
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
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, ksplit, scratch);
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

// the bf16 x 6 kernel takes what the lean kernel takes (32-bit operand offsets) with K a multiple of 32
bool gemm_x6_eligible(const ConvParams& p) {
  return (p.Cin % 32) == 0 && (p.K % 32) == 0 && (p.lda & 3) == 0 && ((uintptr_t)p.in & 15) == 0 &&
         (long)p.B * p.H * p.W * p.lda * 4 < 0x7fffffffL && (long)p.M * p.ldc * 4 < 0x7fffffffL &&
         (!p.res || (long)(p.res_rows ? p.res_rows : p.M) * p.ldr * 4 < 0x7fffffffL);
}

// the member that takes the activations pre-split (p.in = three bf16 planes [3][B*H*W][lda], natural k order; w6 packed in natural
// order too): 64x64 tile, two workgroups per CU
int launch_gemm_x6_presplit(const ConvParams& p, const void* w6n, int cout_pad, hipStream_t s, void* out_planes, int ldp) {
  if (!w6n || (cout_pad % 64) || cout_pad < p.Cout || ((uintptr_t)w6n & 15) || ((uintptr_t)p.in & 15)) return AOT_ERR_UNSUPPORTED;
  if (out_planes && (ldp < p.Cout || (ldp & 7) || 3L * p.M * ldp * 2 >= 0x7fffffffL || ((uintptr_t)out_planes & 15))) return AOT_ERR_BADARG;
  if (!out_planes && !p.out) return AOT_ERR_BADARG;
  if ((p.Cin % 32) || (p.K % 32) || (p.lda & 7)) return AOT_ERR_UNSUPPORTED;
  if (3L * p.B * p.H * p.W * p.lda * 2 >= 0x7fffffffL || (long)p.M * p.ldc * 4 >= 0x7fffffffL ||
      (p.res && (long)(p.res_rows ? p.res_rows : p.M) * p.ldr * 4 >= 0x7fffffffL) || 3L * (p.K / 8) * cout_pad * 16 >= 0x7fffffffL)
    return AOT_ERR_UNSUPPORTED;
  const bool is1x1 = (p.KH == 1 && p.KW == 1 && p.pad == 0);
  X6Weight wq;
  wq.w6 = w6n;
  wq.cout_pad = cout_pad;
  const int nitems = cdiv(p.M, 64) * cdiv(p.Cout, 64);
  const int grid = nitems < 512 ? nitems : 512;
  if (out_planes) {        // the tile end writes planes: their base and row stride ride in the split-K arguments
    if (is1x1)
      hipLaunchKernelGGL((gemm_x6_kernel<true, 6, false, true, true>), dim3(grid), dim3(256), 0, s, p, wq, ldp, (float*)out_planes);
    else
      hipLaunchKernelGGL((gemm_x6_kernel<false, 6, false, true, true>), dim3(grid), dim3(256), 0, s, p, wq, ldp, (float*)out_planes);
    AOT_LAUNCH_CHECK();
  }
  if (is1x1)
    hipLaunchKernelGGL((gemm_x6_kernel<true, 6, false, true>), dim3(grid), dim3(256), 0, s, p, wq, 1, nullptr);
  else
    hipLaunchKernelGGL((gemm_x6_kernel<false, 6, false, true>), dim3(grid), dim3(256), 0, s, p, wq, 1, nullptr);
  AOT_LAUNCH_CHECK();
}

// the phase-shifted 128x128 form (gemm_x6pp_kernel); ksplit > 1: split-K over the grid, slabs [ksplit][M][Cout] in `scratch`
int launch_gemm_x6pp(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s, int ksplit, float* scratch) {
  if (!gemm_x6_eligible(p) || !w6 || (cout_pad % 64) || cout_pad < p.Cout || ((uintptr_t)w6 & 15)) return AOT_ERR_UNSUPPORTED;
  if (3L * (p.K / 8) * cout_pad * 16 >= 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  if (ksplit < 1 || (p.K / BK) % ksplit != 0) return AOT_ERR_BADARG;
  if (ksplit > 1 && (!scratch || (long)p.M * p.Cout * 4 >= 0x7fffffffL)) return AOT_ERR_BADARG;
  const bool is1x1 = (p.KH == 1 && p.KW == 1 && p.pad == 0);
  X6Weight wq;
  wq.w6 = w6;
  wq.cout_pad = cout_pad;
  const int nit = cdiv(p.M, 128) * cdiv(p.Cout, 128) * ksplit;
  const int grid = nit < 256 ? nit : 256;                     // one 8-wave workgroup per CU
  if (ksplit > 1) {
    if (is1x1)
      hipLaunchKernelGGL((gemm_x6pp_kernel<true, true>), dim3(grid), dim3(512), 0, s, p, wq, ksplit, scratch);
    else
      hipLaunchKernelGGL((gemm_x6pp_kernel<false, true>), dim3(grid), dim3(512), 0, s, p, wq, ksplit, scratch);
    const long n = (long)p.M * ((p.Cout + 3) >> 2);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, ksplit, scratch);
    AOT_LAUNCH_CHECK();
  }
  if (is1x1)
    hipLaunchKernelGGL((gemm_x6pp_kernel<true, false>), dim3(grid), dim3(512), 0, s, p, wq, 1, nullptr);
  else
    hipLaunchKernelGGL((gemm_x6pp_kernel<false, false>), dim3(grid), dim3(512), 0, s, p, wq, 1, nullptr);
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
  hipLaunchKernelGGL((gemm_x6rd_kernel<true, false, true>), dim3(grid), dim3(256), 0, s, p, wq, 1, gn_part);
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
  hipLaunchKernelGGL((gemm_x6rd_kernel<false, false, false, true>), dim3(grid), dim3(256), 0, s, p, wq, 1, nullptr);
  AOT_LAUNCH_CHECK();
}

// split-K over the grid on the 64x64 register-staged kernel with direct weight fragments (gemm_x6rd_kernel<., true>)
int launch_gemm_x6rd_splitk(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s, int ksplit, float* scratch) {
  if (!gemm_x6_eligible(p) || !w6 || (cout_pad % 64) || cout_pad < p.Cout || ((uintptr_t)w6 & 15)) return AOT_ERR_UNSUPPORTED;
  if (3L * (p.K / 8) * cout_pad * 16 >= 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  if (ksplit < 2 || (p.K / BK) % ksplit != 0 || !scratch || (long)p.M * p.Cout * 4 >= 0x7fffffffL) return AOT_ERR_BADARG;
  X6Weight wq;
  wq.w6 = w6;
  wq.cout_pad = cout_pad;
  const int nit = cdiv(p.M, 64) * cdiv(p.Cout, 64) * ksplit;
  const int grid = nit < 1024 ? nit : 1024;                  // (the split-K form needs 118 registers: four workgroups per CU)
  if (p.KH == 1 && p.KW == 1 && p.pad == 0)
    hipLaunchKernelGGL((gemm_x6rd_kernel<true, true>), dim3(grid), dim3(256), 0, s, p, wq, ksplit, scratch);
  else
    hipLaunchKernelGGL((gemm_x6rd_kernel<false, true>), dim3(grid), dim3(256), 0, s, p, wq, ksplit, scratch);
  const long n = (long)p.M * ((p.Cout + 3) >> 2);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, ksplit, scratch);
  AOT_LAUNCH_CHECK();
}

int launch_gemm_x6(const ConvParams& p, const void* w6, int cout_pad, int tile, hipStream_t s, int terms, int ksplit, float* scratch) {
  if (!gemm_x6_eligible(p) || !w6 || (cout_pad % 64) || cout_pad < p.Cout || ((uintptr_t)w6 & 15)) return AOT_ERR_UNSUPPORTED;
  if (3L * (p.K / 8) * cout_pad * 16 >= 0x7fffffffL) return AOT_ERR_UNSUPPORTED;
  if (tile == 256) return terms == 6 ? launch_gemm_x6pp(p, w6, cout_pad, s, ksplit, scratch) : AOT_ERR_BADARG;
  if (tile == 65) {             // the register-staged 64x64 form: three workgroups per CU
    if (terms != 6 || ksplit != 1) return AOT_ERR_BADARG;
    X6Weight wr;
    wr.w6 = w6;
    wr.cout_pad = cout_pad;
    const int nit = cdiv(p.M, 64) * cdiv(p.Cout, 64);
    const int gr = nit < 768 ? nit : 768;
    if (p.KH == 1 && p.KW == 1 && p.pad == 0)
      hipLaunchKernelGGL((gemm_x6r_kernel<true, 2, 1>), dim3(gr), dim3(256), 0, s, p, wr);
    else
      hipLaunchKernelGGL((gemm_x6r_kernel<false, 2, 1>), dim3(gr), dim3(256), 0, s, p, wr);
    AOT_LAUNCH_CHECK();
  }
  if (tile == 66) {             // the register-staged 64x64 form with the weight fragments straight from global memory
    if (terms != 6 || ksplit != 1) return AOT_ERR_BADARG;
    X6Weight wr;
    wr.w6 = w6;
    wr.cout_pad = cout_pad;
    const int nit = cdiv(p.M, 64) * cdiv(p.Cout, 64);
    const int gr = nit < 768 ? nit : 768;
    if (p.KH == 1 && p.KW == 1 && p.pad == 0)
      hipLaunchKernelGGL((gemm_x6rd_kernel<true, false>), dim3(gr), dim3(256), 0, s, p, wr, 1, nullptr);
    else
      hipLaunchKernelGGL((gemm_x6rd_kernel<false, false>), dim3(gr), dim3(256), 0, s, p, wr, 1, nullptr);
    AOT_LAUNCH_CHECK();
  }
  if (tile == 129) {            // the register-staged 128x128 form: eight waves, one workgroup per CU
    if (terms != 6 || ksplit != 1) return AOT_ERR_BADARG;
    X6Weight wr;
    wr.w6 = w6;
    wr.cout_pad = cout_pad;
    const int nit = cdiv(p.M, 128) * cdiv(p.Cout, 128);
    const int gr = nit < 256 ? nit : 256;
    if (p.KH == 1 && p.KW == 1 && p.pad == 0)
      hipLaunchKernelGGL((gemm_x6r_kernel<true, 4, 2>), dim3(gr), dim3(512), 0, s, p, wr);
    else
      hipLaunchKernelGGL((gemm_x6r_kernel<false, 4, 2>), dim3(gr), dim3(512), 0, s, p, wr);
    AOT_LAUNCH_CHECK();
  }
  const bool is1x1 = (p.KH == 1 && p.KW == 1 && p.pad == 0);
  X6Weight wq;
  wq.w6 = w6;
  wq.cout_pad = cout_pad;
  if (ksplit < 1 || (ksplit > 1 && terms != 1)) return AOT_ERR_BADARG;
  if (terms == 1) {           // plain bf16 (training): the 64x64 tile, two workgroups per CU
    const int nit = cdiv(p.M, 64) * cdiv(p.Cout, 64);
    const int g1 = nit < 512 ? nit : 512;
    if (ksplit > 1) {         // split-K (weight gradients): 1x1 only, K / 32 divisible, partial slabs + the reduce pass
      if (!is1x1 || (p.K / BK) % ksplit != 0 || !scratch || (long)p.M * p.Cout * 4 >= 0x7fffffffL) return AOT_ERR_BADARG;
      const int gk = nit * ksplit < 512 ? nit * ksplit : 512;
      hipLaunchKernelGGL((gemm_x6_kernel<true, 1, true>), dim3(gk), dim3(256), 0, s, p, wq, ksplit, scratch);
      const long n = (long)p.M * ((p.Cout + 3) >> 2);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, ksplit, scratch);
      AOT_LAUNCH_CHECK();
    }
    if (is1x1)
      hipLaunchKernelGGL((gemm_x6_kernel<true, 1>), dim3(g1), dim3(256), 0, s, p, wq, 1, nullptr);
    else
      hipLaunchKernelGGL((gemm_x6_kernel<false, 1>), dim3(g1), dim3(256), 0, s, p, wq, 1, nullptr);
    AOT_LAUNCH_CHECK();
  }
  const int nwide = cdiv(p.M, 128) * cdiv(p.Cout, 128);
  // Round 5 (profiles/r05_x6r.txt, r05_x6r128.txt: every conv / linear of the frame at batch 1 and 3): the register-staged 64x64
  // kernel (tile 65) is the default of the family -- faster than the LDS-DMA 64x64 kernel on all but two shapes and than the 128x128
  // kernels on every 1x1 layer; the 128x128 tile (its register-staged form, tile 129) keeps the KxK layers that fill the chip with it
  // (>= 200 tiles, rounds >= 70 % full: the 3x3 convolutions of the decoder at the 4x map), whose activation rows it re-reads half as
  // often across the filter taps.  The LDS-DMA kernels (tiles 64 / 128 / 256) stay in the library: tile 1 = the round-4 rule.
  const int rounds = (nwide + 255) / 256;
  // (with the weight fragments straight from global memory -- tile 66, profiles/r05_x6rd.txt -- the 64x64 form also beats the 128x128
  //  one on the 4x map at three lanes, 139.7 vs 146.0 us; the 128x128 tile keeps the KxK layers that fill ONE dispatch round with it)
  const bool wide = tile == 128 || (tile == 0 && p.KH * p.KW > 1 && p.Cout >= 128 && nwide >= 200 && rounds == 1) ||
                    // tile 1 = the round-4 rule (A/B runs: AOT_X6_TILE=1): 128x128 wherever it fills the chip, else the LDS-DMA 64x64 kernel
                    (tile == 1 && p.Cout >= 128 && nwide >= 150 && (rounds == 1 || 10 * nwide >= 7 * 256 * rounds));
  if (tile == 0) {
    // (development switch for A/B runs in one process tree: AOT_X6_DEF64=65 makes the both-operands-staged kernel the 64x64 default)
    static const int def64 = [] { const char* e = getenv("AOT_X6_DEF64"); return (e && atoi(e) == 65) ? 65 : 66; }();
    return launch_gemm_x6(p, w6, cout_pad, wide ? 129 : def64, s, terms, ksplit, scratch);
  }
  if (wide) {
    const int grid = nwide < 256 ? nwide : 256;               // one 8-wave workgroup per CU
    if (is1x1)
      hipLaunchKernelGGL((gemm_x6w_kernel<true>), dim3(grid), dim3(512), 0, s, p, wq);
    else
      hipLaunchKernelGGL((gemm_x6w_kernel<false>), dim3(grid), dim3(512), 0, s, p, wq);
    AOT_LAUNCH_CHECK();
  }
  const int nitems = cdiv(p.M, 64) * cdiv(p.Cout, 64);
  const int grid = nitems < 512 ? nitems : 512;             // two workgroups per CU
  if (is1x1)
    hipLaunchKernelGGL((gemm_x6_kernel<true>), dim3(grid), dim3(256), 0, s, p, wq, 1, nullptr);
  else
    hipLaunchKernelGGL((gemm_x6_kernel<false>), dim3(grid), dim3(256), 0, s, p, wq, 1, nullptr);
  AOT_LAUNCH_CHECK();
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
