// HBM/L2-bound glue kernels of the AOT frame: normalisations, depthwise conv, pooling, resize,
// layout changes, identity-bank gather, logit finalisation.  All are coalesced float4 streamers over
// token-major (NHWC) maps; none of them is reshaped into a GEMM.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// LayerNorm: one 64-lane wave per token row, row kept in registers (C <= 1024), two-pass variance.
// ---------------------------------------------------------------------------------------------
template <int NV>  // float4 per lane
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        const float* __restrict__ add, float* __restrict__ y2, int M,
                                                        int C, int ldx, int ldy, int ldadd, int ldy2, int add_rows,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = C >> 2;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + i * 64;
    v[i] = (c4 < nv) ? *reinterpret_cast<const float4*>(x + (long)row * ldx + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + i * 64;
    if (c4 < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      sq += (a * a + b * b) + (c * c + d * d);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
  const float rstd = 1.f / sqrtf(sq / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + i * 64;
    if (c4 < nv) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c4 * 4);
      const float4 b = *reinterpret_cast<const float4*>(beta + c4 * 4);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      *reinterpret_cast<float4*>(y + (long)row * ldy + c4 * 4) = o;
      if (y2) {
        const float4 a = *reinterpret_cast<const float4*>(add + (long)(add_rows ? row % add_rows : row) * ldadd + c4 * 4);
        *reinterpret_cast<float4*>(y2 + (long)row * ldy2 + c4 * 4) = make_float4(o.x + a.x, o.y + a.y, o.z + a.z, o.w + a.w);
      }
    }
  }
}

extern "C" int aot_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, const float* add,
                                 float* y2, int M, int C, int ldx, int ldy, int ldadd, int ldy2, int add_rows, float eps,
                                 void* stream) {
  if (add_rows < 0) return AOT_ERR_BADARG;
  if (!x || !gamma || !beta || !y || M <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldy & 3)) return AOT_ERR_BADARG;
  if (y2 && (!add || (ldadd & 3) || (ldy2 & 3))) return AOT_ERR_BADARG;
  if (C > 1024) return AOT_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(cdiv(M, 4)), block(256);
  if (C <= 256)
    hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, gamma, beta, y, add, y2, M, C, ldx, ldy, ldadd, ldy2, add_rows, eps);
  else if (C <= 512)
    hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, gamma, beta, y, add, y2, M, C, ldx, ldy, ldadd, ldy2, add_rows, eps);
  else
    hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, x, gamma, beta, y, add, y2, M, C, ldx, ldy, ldadd, ldy2, add_rows, eps);
  AOT_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// GroupNorm over B lanes of [M, C] NHWC maps (lane b = rows [b*M, (b+1)*M)): deterministic two-level reduction in fp64
// in ONE launch -- the last of a (lane, group)'s nsplit partial-sum workgroups to arrive (device-scope ticket) adds the
// partials in index order and writes (mean, rstd) -- then a streaming apply, or the fused apply + GELU + 5x5 depthwise
// conv below.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gn_stats_kernel(const float* __restrict__ x, double* __restrict__ scratch,
                                                       double* __restrict__ stats, unsigned* __restrict__ ticket, int M,
                                                       int C, int G, int ldx, int nsplit, float eps) {
  const int g = blockIdx.y, sp = blockIdx.x, bl = blockIdx.z;
  const int cg = C / G, v4 = cg >> 2;  // float4 per row of this group
  const int rows_per_pass = 256 / v4;
  const int t = threadIdx.x;
  const int r_in = t / v4, c4 = t - r_in * v4;
  const int rows = (M + nsplit - 1) / nsplit;
  const int r0 = sp * rows, r1 = min(M, r0 + rows);
  const float* xb = x + (long)bl * M * ldx;
  double s = 0.0, sq = 0.0;
  if (r_in < rows_per_pass) {
    for (int r = r0 + r_in; r < r1; r += rows_per_pass) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (long)r * ldx + g * cg + c4 * 4);
      s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      sq += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
  }
  __shared__ double red[2][4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off);
    sq += __shfl_xor(sq, off);
  }
  if ((t & 63) == 0) { red[0][t >> 6] = s; red[1][t >> 6] = sq; }
  __syncthreads();
  const long slot = (long)bl * G + g;
  double* part = scratch + slot * nsplit * 2;
  if (t == 0) {
    const double ps = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double pq = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    int is_last = 1;
    if (nsplit > 1) {
      // publish the partial, then take a ticket; the workgroup that draws the last ticket reduces.  Both sides use 8-byte
      // agent-scope atomics (`sc1`: stores write through to L2, loads bypass the CU's L1), so no cache needs flushing or
      // invalidating -- the stores only have to be acknowledged before the ticket goes out (MI355X_MICROARCH.md: "sc1 payload
      // -> asm vmcnt(0) -> flag"; an agent-scope release + acquire fence pair here cost 3-8 us of the launch's 11.5)
      __hip_atomic_store(&part[sp * 2], ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&part[sp * 2 + 1], pq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned prev = __hip_atomic_fetch_add(&ticket[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = prev == (unsigned)(nsplit - 1);
    }
    if (is_last) {
      double ts = 0.0, tq = 0.0;
      if (nsplit > 1) {
        for (int i = 0; i < nsplit; ++i) {   // index order, whoever arrived last: deterministic
          ts += __hip_atomic_load(&part[i * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tq += __hip_atomic_load(&part[i * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(&ticket[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call / replay
      } else {
        ts = ps; tq = pq;
      }
      const double cnt = (double)M * cg;
      const double mean = ts / cnt;
      double var = tq / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      stats[slot * 2] = mean;
      stats[slot * 2 + 1] = 1.0 / sqrt(var + (double)eps);
    }
  }
}

// Round 6: the statistics pass with WHOLE ROWS per workgroup.  gn_stats_kernel gives every (row range, group) its own workgroup, which
// reads one group's 64- or 128-byte piece of every 512-byte row (12 us for the 13 MB stride-4 map of the FPN head: 1.1 TB/s).  Here a
// workgroup owns a row range and ALL groups: thread = (float4 column, row of the pass), fully coalesced rows; the per-thread fp64 partials
// of a group meet in LDS (fixed order), one (sum, sum of squares) pair per (range, group) is published, and the last workgroup of a lane
// to arrive adds the ranges in index order -- the same two-level fp64 reduction and the same ticket protocol as gn_stats_kernel, hence the
// same statistics up to the order of fp64 additions.  Needs C / 4 <= 256 dividing 256 and (C / G) % 4 == 0.
__global__ void __launch_bounds__(256) gn_stats_rows_kernel(const float* __restrict__ x, double* __restrict__ scratch,
                                                            double* __restrict__ stats, unsigned* __restrict__ ticket, int M,
                                                            int C, int G, int ldx, int nsplit, float eps) {
  const int sp = blockIdx.x, bl = blockIdx.y, t = threadIdx.x;
  const int nv = C >> 2, rpp = 256 / nv, v4g = (C / G) >> 2;      // float4 per row, rows per pass, float4 per group and row
  const int c4 = t % nv, rin = t / nv;
  const int rows = (M + nsplit - 1) / nsplit;
  const int r0 = sp * rows, r1 = min(M, r0 + rows);
  const float* xb = x + (long)bl * M * ldx;
  double s = 0.0, sq = 0.0;
  for (int r = r0 + rin; r < r1; r += rpp) {
    const float4 v = *reinterpret_cast<const float4*>(xb + (long)r * ldx + c4 * 4);
    s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    sq += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
  }
  __shared__ double red[256][2];
  __shared__ int last_flag;
  red[t][0] = s;
  red[t][1] = sq;
  __syncthreads();
  double* part = scratch + (long)bl * G * nsplit * 2;          // [G][nsplit][2]
  if (t < 64) {          // (one wave: its stores are one instruction each, the wait below covers all of them)
    if (t < G) {
      double ps = 0.0, pq = 0.0;
      for (int ri = 0; ri < rpp; ++ri)
        for (int c = 0; c < v4g; ++c) {
          ps += red[ri * nv + t * v4g + c][0];
          pq += red[ri * nv + t * v4g + c][1];
        }
      __hip_atomic_store(&part[((long)t * nsplit + sp) * 2], ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&part[((long)t * nsplit + sp) * 2 + 1], pq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t == 0) {
      const unsigned prev = nsplit > 1 ? __hip_atomic_fetch_add(&ticket[(long)bl * G], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      last_flag = prev == (unsigned)(nsplit - 1);
    }
  }
  __syncthreads();
  if (!last_flag) return;
  // the last workgroup of the lane: thread = (group t % G, strand t / G); strand i adds ranges i, i + 256 / G, ... in index order, then thread
  // g < G adds its group's strands in strand order
  {
    const int g = t % G, strand = t / G, nstr = 256 / G;
    double ts = 0.0, tq = 0.0;
    for (int i = strand; i < nsplit; i += nstr) {
      ts += __hip_atomic_load(&part[((long)g * nsplit + i) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tq += __hip_atomic_load(&part[((long)g * nsplit + i) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    red[t][0] = ts;
    red[t][1] = tq;
    __syncthreads();
    if (t < G) {
      double as = 0.0, aq = 0.0;
      for (int i = 0; i < nstr; ++i) { as += red[i * G + t][0]; aq += red[i * G + t][1]; }
      const double cnt = (double)M * (C / G);
      const double mean = as / cnt;
      double var = aq / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      stats[((long)bl * G + t) * 2] = mean;
      stats[((long)bl * G + t) * 2 + 1] = 1.0 / sqrt(var + (double)eps);
    }
    if (t == 0 && nsplit > 1) __hip_atomic_store(&ticket[(long)bl * G], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ float gn_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 3) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));  // exact-erf GELU (F.gelu default)
  return v;
}

__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, int M, int C, int G, int ldx, int ldy,
                                                       int act, long total, const float* __restrict__ add, int ldadd,
                                                       int add_rows) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nv = C >> 2;
  if (idx >= total) return;
  const int c4 = (int)(idx % nv);
  const long r = idx / nv;                 // row over all lanes
  const int bl = (int)(r / M);
  const int g = (c4 * 4) / (C / G);
  const float mean = (float)stats[((long)bl * G + g) * 2], rstd = (float)stats[((long)bl * G + g) * 2 + 1];
  const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c4 * 4);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c4 * 4);
  const float4 be = *reinterpret_cast<const float4*>(beta + c4 * 4);
  float4 o;
  o.x = gn_act((v.x - mean) * rstd * ga.x + be.x, act);
  o.y = gn_act((v.y - mean) * rstd * ga.y + be.y, act);
  o.z = gn_act((v.z - mean) * rstd * ga.z + be.z, act);
  o.w = gn_act((v.w - mean) * rstd * ga.w + be.w, act);
  if (add) {     // + a map added after the activation (shared by the lanes when add_rows > 0): FPN shortcut adapters
    const float4 e = *reinterpret_cast<const float4*>(add + (add_rows ? r % add_rows : r) * ldadd + c4 * 4);
    o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
  }
  *reinterpret_cast<float4*>(y + r * ldy + c4 * 4) = o;
}

extern "C" int aot_groupnorm_stats_f32(const float* x, double* scratch, double* stats, unsigned* ticket, int B, int M,
                                       int C, int G, int ldx, float eps, int nsplit, void* stream) {
  if (!x || !scratch || !stats || B <= 0 || M <= 0 || C <= 0 || G <= 0 || C % G || ((C / G) & 3) || (ldx & 3) || nsplit < 1)
    return AOT_ERR_BADARG;
  if (nsplit > 1 && !ticket) return AOT_ERR_BADARG;
  if ((C / G) / 4 > 256 || B > 65535) return AOT_ERR_UNSUPPORTED;
  const int nv = C / 4;
  if (nv <= 256 && 256 % nv == 0 && ((C / G) & 3) == 0 && 256 % G == 0 && G <= 64 && nsplit >= 64) {
    // whole rows per workgroup (round 6): the callers that ask for many row ranges (the FPN head's maps) get the coalesced form
    hipLaunchKernelGGL(gn_stats_rows_kernel, dim3(nsplit, B), dim3(256), 0, (hipStream_t)stream, x, scratch, stats, ticket, M, C, G, ldx,
                       nsplit, eps);
    AOT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nsplit, G, B), dim3(256), 0, (hipStream_t)stream, x, scratch, stats, ticket, M, C,
                     G, ldx, nsplit, eps);
  AOT_LAUNCH_CHECK();
}

extern "C" int aot_groupnorm_apply_f32(const float* x, const double* stats, const float* gamma, const float* beta,
                                       float* y, const float* add, int B, int M, int C, int G, int ldx, int ldy, int ldadd,
                                       int add_rows, int act, void* stream) {
  if (!x || !stats || !gamma || !beta || !y || B <= 0 || M <= 0 || C <= 0 || G <= 0 || C % G || ((C / G) & 3) ||
      (ldx & 3) || (ldy & 3))
    return AOT_ERR_BADARG;
  if (add && ((ldadd & 3) || ldadd < C || add_rows < 0)) return AOT_ERR_BADARG;
  const long total = (long)B * M * (C / 4);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, stats, gamma, beta, y,
                     M, C, G, ldx, ldy, act, total, add, ldadd, add_rows);
  AOT_LAUNCH_CHECK();
}

// Fused GroupNorm-apply + activation + 5x5 depthwise conv (stride 1, pad 2) for 32-channel groups: GNActDWConv2d
// (basic.py:15-35) after the statistics pass, i.e. gn -> GELU -> conv in one launch.  A workgroup owns an 8x8 output
// tile of one (lane, group): the 12x12 input halo is normalised and activated ONCE per element on its way into LDS (the
// conv zero-pads AFTER the activation), then every thread (pixel, 8 channels) walks the 25 taps out of LDS.
// PART (round 5): the statistics arrive as the PARTIAL sums the producing GEMM's tile end wrote (aot_linear_gn_bf16x6_f32:
// part[P][G][2] floats, one (sum, squared deviations) per 32-row block and group; one lane): every wave adds its group's P partials
// in double -- lane l takes partials l, l + 64, ... in index order, then a fixed butterfly over the 64 lanes: the same bits in every
// wave, workgroup and run -- and forms (mean, rstd) as gn_stats_kernel does.  The statistics launch and its pass over the map
// are gone.
// PLAIN (round 6): no normalisation, no activation -- the tiled 5x5 depthwise convolution on its own (the dw_conv of the GPM blocks' tails,
// attention.py:709-710): the 12x12 halo of an 8x8 tile goes through LDS once instead of 25 reads of every element from L2; the taps are
// added in dwconv_kernel's order (zero-padded taps contribute exact zeros): bit-identical to it.
template <bool PART, bool PLAIN = false>
__global__ void __launch_bounds__(256, 4) gn_act_dwconv5_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ w, float* __restrict__ out, int H,
                                                             int W, int C, int G, int ldx, int ldo, int act, int tiles_x,
                                                             const float* __restrict__ part, int P, float eps) {
  constexpr int TH = 8, TW = 8, R = 2, IH = TH + 2 * R, IW = TW + 2 * R, CB = 32;
  __shared__ __attribute__((aligned(16))) float tile[IH * IW][CB + 4];    // +4: rows 36 floats apart (bank spread)
  __shared__ __attribute__((aligned(16))) float wk[25][CB];
  const int g = blockIdx.y, bl = blockIdx.z;
  const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
  const int t = threadIdx.x;
  float mean = 0.f, rstd = 1.f;
  if (PLAIN) {
  } else if (PART) {
    // partial i = (sum, sum of squared deviations from its own mean) of rows [32 i, 32 i + 32) x this group's CB channels, written by
    // the producing GEMM's tile end; combined with Chan's formula in double, in index order (fixed tree): first the grand mean, then
    // M2 = sum_i [M2_i + n_i (mean_i - mean)^2]
    // (round 6: every WAVE adds the partials on its own -- lane l takes partials l, l + 64, ... in index order, then one butterfly over
    //  the 64 lanes: the same bits in every wave, workgroup and run, and no workgroup barrier; the 256-thread tree of round 5 spent
    //  sixteen barriers on 54 numbers)
    const long rows = (long)H * W;
    const int ln = t & 63;
    double ps = 0.0;
    for (int i = ln; i < P; i += 64) ps += (double)part[((long)i * G + g) * 2];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ps += __shfl_xor(ps, off);
    const double cnt = (double)rows * CB;
    const double m = ps / cnt;
    double pq = 0.0;
    for (int i = ln; i < P; i += 64) {
      const long nr = min(32L, max(0L, rows - 32L * i));
      if (nr > 0) {
        const double ni = (double)nr * CB, di = (double)part[((long)i * G + g) * 2] / ni - m;
        pq += (double)part[((long)i * G + g) * 2 + 1] + ni * di * di;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pq += __shfl_xor(pq, off);
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(pq / cnt + (double)eps));
  } else {
    mean = (float)stats[((long)bl * G + g) * 2];
    rstd = (float)stats[((long)bl * G + g) * 2 + 1];
  }
  const float* xb = x + (long)bl * H * W * ldx + g * CB;
  const int c4 = t & 7;                          // channel quad of the group
  const float4 ga = PLAIN ? make_float4(1.f, 1.f, 1.f, 1.f) : *reinterpret_cast<const float4*>(gamma + g * CB + c4 * 4);
  const float4 be = PLAIN ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(beta + g * CB + c4 * 4);
  for (int i = t >> 3; i < IH * IW; i += 32) {
    const int iy = ty0 - R + i / IW, ix = tx0 - R + i % IW;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
      const float4 v = *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx + c4 * 4);
      if (PLAIN) {
        o = v;
      } else {
        o.x = gn_act((v.x - mean) * rstd * ga.x + be.x, act);
        o.y = gn_act((v.y - mean) * rstd * ga.y + be.y, act);
        o.z = gn_act((v.z - mean) * rstd * ga.z + be.z, act);
        o.w = gn_act((v.w - mean) * rstd * ga.w + be.w, act);
      }
    }
    *reinterpret_cast<float4*>(&tile[i][c4 * 4]) = o;
  }
  for (int i = t; i < 25 * (CB / 4); i += 256)
    *reinterpret_cast<float4*>(&wk[i >> 3][(i & 7) * 4]) = *reinterpret_cast<const float4*>(w + (long)(i >> 3) * C + g * CB + (i & 7) * 4);
  __syncthreads();
  // thread = (pixel of the 8x8 tile, channel octet): 64 x 4
  const int pix = t >> 2, co = (t & 3) * 8;
  const int py = pix >> 3, px = pix & 7;
  const int oy = ty0 + py, ox = tx0 + px;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ky = 0; ky < 5; ++ky)
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
      const float* src = &tile[(py + ky) * IW + px + kx][co];
      const float* kk = &wk[ky * 5 + kx][co];
      const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
      const float4 k0 = *reinterpret_cast<const float4*>(kk), k1 = *reinterpret_cast<const float4*>(kk + 4);
      acc[0] = fmaf(a0.x, k0.x, acc[0]); acc[1] = fmaf(a0.y, k0.y, acc[1]);
      acc[2] = fmaf(a0.z, k0.z, acc[2]); acc[3] = fmaf(a0.w, k0.w, acc[3]);
      acc[4] = fmaf(a1.x, k1.x, acc[4]); acc[5] = fmaf(a1.y, k1.y, acc[5]);
      acc[6] = fmaf(a1.z, k1.z, acc[6]); acc[7] = fmaf(a1.w, k1.w, acc[7]);
    }
  if (oy < H && ox < W) {
    float* dst = out + ((long)bl * H * W + (long)oy * W + ox) * ldo + g * CB + co;
    *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

extern "C" int aot_gn_act_dwconv5_f32(const float* x, const double* stats, const float* gamma, const float* beta,
                                      const float* w, float* out, int B, int H, int W, int C, int G, int ldx, int ldo,
                                      int act, void* stream) {
  if (!x || !stats || !gamma || !beta || !w || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || G <= 0 || (ldx & 3) || (ldo & 3))
    return AOT_ERR_BADARG;
  if (C != G * 32 || B > 65535 || G > 65535) return AOT_ERR_UNSUPPORTED;
  const int tx = cdiv(W, 8), ty = cdiv(H, 8);
  hipLaunchKernelGGL(gn_act_dwconv5_kernel<false>, dim3(tx * ty, G, B), dim3(256), 0, (hipStream_t)stream, x, stats, gamma, beta, w,
                     out, H, W, C, G, ldx, ldo, act, tx, (const float*)nullptr, 0, 0.f);
  AOT_LAUNCH_CHECK();
}

// the same with the statistics taken from the producing GEMM's partial sums (one lane): part [P][G][2] floats
extern "C" int aot_gn_act_dwconv5p_f32(const float* x, const float* part, int P, const float* gamma, const float* beta,
                                       const float* w, float* out, int H, int W, int C, int G, int ldx, int ldo, int act,
                                       float eps, void* stream) {
  if (!x || !part || P <= 0 || !gamma || !beta || !w || !out || H <= 0 || W <= 0 || C <= 0 || G <= 0 || (ldx & 3) || (ldo & 3))
    return AOT_ERR_BADARG;
  if (C != G * 32 || G > 65535) return AOT_ERR_UNSUPPORTED;
  const int tx = cdiv(W, 8), ty = cdiv(H, 8);
  hipLaunchKernelGGL(gn_act_dwconv5_kernel<true>, dim3(tx * ty, G, 1), dim3(256), 0, (hipStream_t)stream, x, (const double*)nullptr,
                     gamma, beta, w, out, H, W, C, G, ldx, ldo, act, tx, part, P, eps);
  AOT_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Depthwise KxK conv, NHWC: thread = (output pixel, 4 channels); taps are coalesced float4 rows.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dwconv_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                     int C, int OH, int OW, int KH, int KW, int stride, int pad, int dil,
                                                     int act) {
  in += (long)blockIdx.y * H * W * C;          // lane of the batch
  out += (long)blockIdx.y * OH * OW * C;
  // XCD-aware block order: block b runs on XCD b%8; give each XCD a contiguous range of pixels so the KxK
  // neighbourhoods it re-reads stay in its own L2 (PMC: 62 MB fetched for a 7 MB map with the plain order)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, sub = bid >> 3;
  const int blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + sub;
  const long idx = (long)blk * blockDim.x + threadIdx.x;
  const int nv = C >> 2;
  if (idx >= (long)OH * OW * nv) return;
  const int c4 = (int)(idx % nv);
  const int pix = (int)(idx / nv);
  const int oy = pix / OW, ox = pix - oy * OW;
  float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ky = 0; ky < KH; ++ky) {
    const int iy = oy * stride - pad + ky * dil;
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int kx = 0; kx < KW; ++kx) {
      const int ix = ox * stride - pad + kx * dil;
      if ((unsigned)ix >= (unsigned)W) continue;
      const float4 v = *reinterpret_cast<const float4*>(in + ((long)iy * W + ix) * C + c4 * 4);
      const float4 k = *reinterpret_cast<const float4*>(w + (long)(ky * KW + kx) * C + c4 * 4);
      acc.x = fmaf(v.x, k.x, acc.x);
      acc.y = fmaf(v.y, k.y, acc.y);
      acc.z = fmaf(v.z, k.z, acc.z);
      acc.w = fmaf(v.w, k.w, acc.w);
    }
  }
  acc.x = apply_act(acc.x, act); acc.y = apply_act(acc.y, act);
  acc.z = apply_act(acc.z, act); acc.w = apply_act(acc.w, act);
  *reinterpret_cast<float4*>(out + (long)pix * C + c4 * 4) = acc;
}

extern "C" int aot_dwconv2d_nhwc_f32(const float* in, const float* w, const float* bias, float* out, int B, int H, int W,
                                     int C, int OH, int OW, int KH, int KW, int stride, int pad, int dil, int act,
                                     void* stream) {
  if (!in || !w || !out || B <= 0 || B > 65535 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || OH <= 0 || OW <= 0) return AOT_ERR_BADARG;
  if (KH == 5 && KW == 5 && stride == 1 && pad == 2 && dil == 1 && OH == H && OW == W && !bias && act == 0 && (C & 31) == 0 && C / 32 <= 65535) {
    // the LDS-tiled form (gn_act_dwconv5_kernel<false, true>): bit-identical to dwconv_kernel (profiles/r06_dwconv_tiled.txt: +2 % on R50-DeAOTL)
    const int tx = cdiv(W, 8), ty = cdiv(H, 8);
    hipLaunchKernelGGL((gn_act_dwconv5_kernel<false, true>), dim3(tx * ty, C / 32, B), dim3(256), 0, (hipStream_t)stream, in, (const double*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, w, out, H, W, C, C / 32, C, C, 0, tx, (const float*)nullptr, 0, 0.f);
    AOT_LAUNCH_CHECK();
  }
  const long total = (long)OH * OW * (C / 4);
  hipLaunchKernelGGL(dwconv_kernel, dim3(cdiv(total, 256), B), dim3(256), 0, (hipStream_t)stream, in, w, bias, out, H, W, C,
                     OH, OW, KH, KW, stride, pad, dil, act);
  AOT_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                      int C, int OH, int OW) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nv = C >> 2;
  if (idx >= (long)OH * OW * nv) return;
  const int c4 = (int)(idx % nv);
  const int pix = (int)(idx / nv);
  const int oy = pix / OW, ox = pix - oy * OW;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 - 1 + kx;
      if ((unsigned)ix >= (unsigned)W) continue;
      const float4 v = *reinterpret_cast<const float4*>(in + ((long)iy * W + ix) * C + c4 * 4);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  }
  *reinterpret_cast<float4*>(out + (long)pix * C + c4 * 4) = m;
}

extern "C" int aot_maxpool3x3s2_nhwc_f32(const float* in, float* out, int H, int W, int C, int OH, int OW, void* stream) {
  if (!in || !out || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return AOT_ERR_BADARG;
  const long total = (long)OH * OW * (C / 4);
  hipLaunchKernelGGL(maxpool_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, H, W, C, OH, OW);
  AOT_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                           long HW, int Cpad) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  for (int c = 0; c < Cpad; ++c) out[p * Cpad + c] = (c < C) ? in[(long)c * HW + p] : 0.f;
}

extern "C" int aot_nchw_to_nhwc_f32(const float* in, float* out, int C, int H, int W, int Cpad, void* stream) {
  if (!in || !out || C <= 0 || Cpad < C) return AOT_ERR_BADARG;
  const long HW = (long)H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, (hipStream_t)stream, in, out, C, HW, Cpad);
  AOT_LAUNCH_CHECK();
}

__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                           long HW, int ld) {
  // 64 pixels x 64 channels tile through LDS so both sides are coalesced
  __shared__ float tile[64][65];
  const long p0 = (long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const long p = p0 + i;
    const int c = c0 + tx;
    tile[i][tx] = (p < HW && c < C) ? in[p * ld + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i;
    const long p = p0 + tx;
    if (c < C && p < HW) out[(long)c * HW + p] = tile[tx][i];
  }
}

extern "C" int aot_nhwc_to_nchw_f32(const float* in, float* out, int C, int H, int W, int ld, void* stream) {
  if (!in || !out || C <= 0 || ld < C) return AOT_ERR_BADARG;
  const long HW = (long)H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(HW, 64), cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, in, out, C,
                     HW, ld);
  AOT_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Bilinear resize with torch's fp32 source-index arithmetic (ATen UpSample.h:
// area_pixel_compute_scale / area_pixel_compute_source_index / guard_index_and_lambda).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bilinear_coord(int dst, int in_size, int out_size, float scale, int align, int& i0, int& i1,
                                               float& w0, float& w1) {
  if (in_size == out_size) { i0 = i1 = dst; w0 = 1.f; w1 = 0.f; return; }
  float src;
  if (align) src = scale * (float)dst;
  else {
    src = fmaf(scale, (float)dst + 0.5f, -0.5f);   // torch (CPU AVX2 and CUDA builds) contracts this into one FMA
    if (src < 0.f) src = 0.f;
  }
  i0 = min((int)floorf(src), in_size - 1);
  w1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
  w0 = 1.f - w1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
}

static inline float bilinear_scale(int in_size, int out_size, int align) {
  if (align) return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  return (float)in_size / (float)out_size;
}

// GN (round 6): the input is read THROUGH GroupNorm-apply + activation -- in = the un-normalised map of a ConvGN block, stats its
// finished (mean, rstd) per lane and group: the four taps are normalised on the fly with gn_apply_kernel's own arithmetic (same
// order, fp32), so the result is bit-identical to gn_apply -> bilinear while the normalised map is never written (the FPN head's
// conv_16x / conv_8x outputs feed nothing but the next upsampling: fpn.py:40-44, 50-51 in the reference).
template <bool GN>
__global__ void __launch_bounds__(256) bilinear_kernel(const float* __restrict__ in, const float* __restrict__ add,
                                                       float* __restrict__ out, int IH, int IW, int OH, int OW, int C,
                                                       int ldi, int ldadd, int ldo, int align, float sh, float sw,
                                                       int add_shared, const double* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int G, int act) {
  in += (long)blockIdx.y * IH * IW * ldi;       // lane of the batch; `add` is one map per lane, or one shared by all lanes
  out += (long)blockIdx.y * OH * OW * ldo;
  if (add && !add_shared) add += (long)blockIdx.y * OH * OW * ldadd;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nv = C >> 2;
  if (idx >= (long)OH * OW * nv) return;
  const int c4 = (int)(idx % nv);
  const int pix = (int)(idx / nv);
  const int oy = pix / OW, ox = pix - oy * OW;
  int y0, y1, x0, x1;
  float wy0, wy1, wx0, wx1;
  bilinear_coord(oy, IH, OH, sh, align, y0, y1, wy0, wy1);
  bilinear_coord(ox, IW, OW, sw, align, x0, x1, wx0, wx1);
  float4 a = *reinterpret_cast<const float4*>(in + ((long)y0 * IW + x0) * ldi + c4 * 4);
  float4 b = *reinterpret_cast<const float4*>(in + ((long)y0 * IW + x1) * ldi + c4 * 4);
  float4 c = *reinterpret_cast<const float4*>(in + ((long)y1 * IW + x0) * ldi + c4 * 4);
  float4 d = *reinterpret_cast<const float4*>(in + ((long)y1 * IW + x1) * ldi + c4 * 4);
  if (GN) {
    const int g = (c4 * 4) / (C / G);
    const double* st = stats + ((long)blockIdx.y * G + g) * 2;
    const float mean = (float)st[0], rstd = (float)st[1];
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c4 * 4);
    const float4 be = *reinterpret_cast<const float4*>(beta + c4 * 4);
    auto norm = [&](float4& v) {
      v.x = gn_act((v.x - mean) * rstd * ga.x + be.x, act);
      v.y = gn_act((v.y - mean) * rstd * ga.y + be.y, act);
      v.z = gn_act((v.z - mean) * rstd * ga.z + be.z, act);
      v.w = gn_act((v.w - mean) * rstd * ga.w + be.w, act);
    };
    norm(a); norm(b); norm(c); norm(d);
  }
  float4 o;
  o.x = wy0 * (wx0 * a.x + wx1 * b.x) + wy1 * (wx0 * c.x + wx1 * d.x);
  o.y = wy0 * (wx0 * a.y + wx1 * b.y) + wy1 * (wx0 * c.y + wx1 * d.y);
  o.z = wy0 * (wx0 * a.z + wx1 * b.z) + wy1 * (wx0 * c.z + wx1 * d.z);
  o.w = wy0 * (wx0 * a.w + wx1 * b.w) + wy1 * (wx0 * c.w + wx1 * d.w);
  if (add) {
    const float4 e = *reinterpret_cast<const float4*>(add + (long)pix * ldadd + c4 * 4);
    o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
  }
  *reinterpret_cast<float4*>(out + (long)pix * ldo + c4 * 4) = o;
}

extern "C" int aot_bilinear_nhwc_f32(const float* in, const float* add, float* out, int B, int IH, int IW, int OH, int OW,
                                     int C, int ldi, int ldadd, int ldo, int align_corners, int add_shared, void* stream) {
  if (!in || !out || B <= 0 || B > 65535 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0 || (C & 3) || (ldi & 3) ||
      (ldo & 3))
    return AOT_ERR_BADARG;
  if (add && (ldadd & 3)) return AOT_ERR_BADARG;
  const long total = (long)OH * OW * (C / 4);
  hipLaunchKernelGGL(bilinear_kernel<false>, dim3(cdiv(total, 256), B), dim3(256), 0, (hipStream_t)stream, in, add, out, IH, IW, OH,
                     OW, C, ldi, ldadd, ldo, align_corners, bilinear_scale(IH, OH, align_corners),
                     bilinear_scale(IW, OW, align_corners), add_shared, nullptr, nullptr, nullptr, 1, 0);
  AOT_LAUNCH_CHECK();
}

// out = bilinear(act(GroupNorm(in))) (+ add): aot_groupnorm_apply_f32 + aot_bilinear_nhwc_f32 in one launch, bit-identical to the pair;
// stats [B][G][2] doubles (mean, rstd) from aot_groupnorm_stats_f32
extern "C" int aot_gn_bilinear_nhwc_f32(const float* in, const double* stats, const float* gamma, const float* beta, const float* add,
                                        float* out, int B, int IH, int IW, int OH, int OW, int C, int G, int ldi, int ldadd, int ldo,
                                        int align_corners, int add_shared, int act, void* stream) {
  if (!in || !stats || !gamma || !beta || !out || B <= 0 || B > 65535 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0 ||
      (C & 3) || (ldi & 3) || (ldo & 3) || G <= 0 || C % G || ((C / G) & 3))
    return AOT_ERR_BADARG;
  if (add && (ldadd & 3)) return AOT_ERR_BADARG;
  const long total = (long)OH * OW * (C / 4);
  hipLaunchKernelGGL(bilinear_kernel<true>, dim3(cdiv(total, 256), B), dim3(256), 0, (hipStream_t)stream, in, add, out, IH, IW, OH,
                     OW, C, ldi, ldadd, ldo, align_corners, bilinear_scale(IH, OH, align_corners),
                     bilinear_scale(IW, OW, align_corners), add_shared, stats, gamma, beta, G, act);
  AOT_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Logit finalisation (aot_engine.py:367-378) for G object groups (lanes) of one frame: per group mask the unused
// identities to -1e10, planar copy at stride 4 (pred_id_logits), bilinear resize to the output size.  G = 1: written
// planar [C, OH, OW] (what the caller's softmax/argmax reads).  G > 1: the groups' resized logits are merged in the same
// kernel by the reference's soft aggregation (AOTInferEngine.soft_logit_aggregation, aot_engine.py:565-582): softmax per
// group, background = product of the groups' background probabilities, clamp to [1e-5, 1 - 1e-5], logit -- output
// [1 + G*(C-1), OH, OW].  Group g holds objects g*(C-1)+1 .. min((g+1)*(C-1), obj_total).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int group_obj_num(int g, int C, int obj_total) { return max(0, min(C - 1, obj_total - g * (C - 1))); }

__global__ void __launch_bounds__(256) logits_planar_kernel(const float* __restrict__ in, float* __restrict__ out4, int HW,
                                                            int C, int ldi, int obj_total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)HW * C) return;
  const int g = blockIdx.y;
  const int c = (int)(idx / HW);
  const int p = (int)(idx - (long)c * HW);
  out4[(long)g * HW * C + idx] = (c > group_obj_num(g, C, obj_total)) ? -1e10f : in[((long)g * HW + p) * ldi + c];
}

template <int MAXC, bool MULTI>      // MULTI = several object groups (soft aggregation); the single-group form is a plain resize
__global__ void __launch_bounds__(256) logits_resize_kernel(const float* __restrict__ in, float* __restrict__ out, int IH,
                                                            int IW, int C, int ldi, int OH, int OW, int G, int obj_total,
                                                            int align, float sh, float sw, int vec) {
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (long)OH * OW) return;
  const int oy = (int)(pix / OW), ox = (int)(pix - (long)oy * OW);
  int y0, y1, x0, x1;
  float wy0, wy1, wx0, wx1;
  bilinear_coord(oy, IH, OH, sh, align, y0, y1, wy0, wy1);
  bilinear_coord(ox, IW, OW, sw, align, x0, x1, wx0, wx1);
  const long plane = (long)OH * OW;
  float bg = 1.f;
  for (int g = 0; g < G; ++g) {
    const float* base = in + (long)g * IH * IW * ldi;
    const float* pa = base + ((long)y0 * IW + x0) * ldi;
    const float* pb = base + ((long)y0 * IW + x1) * ldi;
    const float* pc = base + ((long)y1 * IW + x0) * ldi;
    const float* pd = base + ((long)y1 * IW + x1) * ldi;
    const int on = group_obj_num(g, C, obj_total);
    float v[MAXC];
    float mx = -INFINITY;
    // the four corner rows as 16-byte loads when the rows allow it (a token row of the decoder output is ldi = 12 floats for
    // 11 logits): 12 instead of 44 load instructions per output pixel
#pragma unroll
    for (int q = 0; q < MAXC / 4; ++q) {
      if (4 * q < C) {
        float ca[4], cb[4], cq[4], cd[4];
        if (vec) {
          const float4 ta = reinterpret_cast<const float4*>(pa)[q], tb = reinterpret_cast<const float4*>(pb)[q];
          const float4 tc = reinterpret_cast<const float4*>(pc)[q], td = reinterpret_cast<const float4*>(pd)[q];
          ca[0] = ta.x; ca[1] = ta.y; ca[2] = ta.z; ca[3] = ta.w;
          cb[0] = tb.x; cb[1] = tb.y; cb[2] = tb.z; cb[3] = tb.w;
          cq[0] = tc.x; cq[1] = tc.y; cq[2] = tc.z; cq[3] = tc.w;
          cd[0] = td.x; cd[1] = td.y; cd[2] = td.z; cd[3] = td.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = min(4 * q + e, C - 1);
            ca[e] = pa[c]; cb[e] = pb[c]; cq[e] = pc[c]; cd[e] = pd[c];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * q + e;
          if (c < C) {
            const bool off = c > on;
            const float a = off ? -1e10f : ca[e], b = off ? -1e10f : cb[e], cc = off ? -1e10f : cq[e], d = off ? -1e10f : cd[e];
            v[c] = wy0 * (wx0 * a + wx1 * b) + wy1 * (wx0 * cc + wx1 * d);
            mx = fmaxf(mx, v[c]);
          }
        }
      }
    }
    if (!MULTI) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) out[(long)c * plane + pix] = v[c];
      return;
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { v[c] = expf(v[c] - mx); sum += v[c]; }
    bg *= v[0] / sum;
#pragma unroll
    for (int c = 1; c < MAXC; ++c)
      if (c < C) {
        const float pr = fminf(fmaxf(v[c] / sum, 1e-5f), 1.f - 1e-5f);
        out[(long)(1 + g * (C - 1) + c - 1) * plane + pix] = logf(pr / (1.f - pr));
      }
  }
  const float pr = fminf(fmaxf(bg, 1e-5f), 1.f - 1e-5f);
  out[pix] = logf(pr / (1.f - pr));
}

extern "C" int aot_logits_finalize_f32(const float* logits, float* out4, float* out, int G, int IH, int IW, int C, int ldi,
                                       int OH, int OW, int obj_total, int align_corners, void* stream) {
  if (!logits || G <= 0 || G > 65535 || IH <= 0 || IW <= 0 || C <= 1 || ldi < C) return AOT_ERR_BADARG;
  if (C > 16) return AOT_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (out4) {
    const long total = (long)IH * IW * C;
    hipLaunchKernelGGL(logits_planar_kernel, dim3(cdiv(total, 256), G), dim3(256), 0, s, logits, out4, IH * IW, C, ldi, obj_total);
  }
  if (out) {
    if (OH <= 0 || OW <= 0) return AOT_ERR_BADARG;
    const int vec = (int)((ldi & 3) == 0 && ((uintptr_t)logits & 15) == 0 && ldi >= ((C + 3) & ~3));
    if (G == 1)
      hipLaunchKernelGGL((logits_resize_kernel<16, false>), dim3(cdiv((long)OH * OW, 256)), dim3(256), 0, s, logits, out, IH, IW,
                         C, ldi, OH, OW, G, obj_total, align_corners, bilinear_scale(IH, OH, align_corners),
                         bilinear_scale(IW, OW, align_corners), vec);
    else
      hipLaunchKernelGGL((logits_resize_kernel<16, true>), dim3(cdiv((long)OH * OW, 256)), dim3(256), 0, s, logits, out, IH, IW,
                         C, ldi, OH, OW, G, obj_total, align_corners, bilinear_scale(IH, OH, align_corners),
                         bilinear_scale(IW, OW, align_corners), vec);
  }
  AOT_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Frame tail in ONE launch (one object group, one augmentation): what aot_logits_finalize_f32 -> aot_fuse_probs_f32 ->
// aot_label_resize_f32 compute in three, without the output-size logits ever touching memory (11 planes of 480 x 854 floats
// written and read again per frame).  Role by thread index: [0, OH*OW): the label of an output pixel -- bilinear resize of the
// masked stride-4 logits (the arithmetic of logits_resize_kernel), softmax, first maximum (the arithmetic of fuse_probs_kernel);
// [OH*OW, + LH*LW): a pixel of the label map fed back to the memory update, the nearest-neighbour pick of label_resize_kernel --
// recomputed from the logits of its source pixel, which gives the SAME value (one deterministic function of the pixel);
// then IH*IW*C threads for the planar stride-4 copy (pred_id_logits).  Bit-identical to the three-launch path.
// Replaces aot_engine.py:367-378 + evaluator.py:332-352,375-381 for the common case.
// ---------------------------------------------------------------------------------------------
template <int MAXC>
__device__ __forceinline__ float tail_label_at(const float* __restrict__ in, int oy, int ox, int IH, int IW, int C, int ldi, int OH,
                                               int OW, int on, int align, float sh, float sw, int vec) {
  int y0, y1, x0, x1;
  float wy0, wy1, wx0, wx1;
  bilinear_coord(oy, IH, OH, sh, align, y0, y1, wy0, wy1);
  bilinear_coord(ox, IW, OW, sw, align, x0, x1, wx0, wx1);
  const float* pa = in + ((long)y0 * IW + x0) * ldi;
  const float* pb = in + ((long)y0 * IW + x1) * ldi;
  const float* pc = in + ((long)y1 * IW + x0) * ldi;
  const float* pd = in + ((long)y1 * IW + x1) * ldi;
  float v[MAXC];
  float m = -INFINITY;
#pragma unroll
  for (int q = 0; q < MAXC / 4; ++q) {
    if (4 * q < C) {
      float ca[4], cb[4], cq[4], cd[4];
      if (vec) {
        const float4 ta = reinterpret_cast<const float4*>(pa)[q], tb = reinterpret_cast<const float4*>(pb)[q];
        const float4 tc = reinterpret_cast<const float4*>(pc)[q], td = reinterpret_cast<const float4*>(pd)[q];
        ca[0] = ta.x; ca[1] = ta.y; ca[2] = ta.z; ca[3] = ta.w;
        cb[0] = tb.x; cb[1] = tb.y; cb[2] = tb.z; cb[3] = tb.w;
        cq[0] = tc.x; cq[1] = tc.y; cq[2] = tc.z; cq[3] = tc.w;
        cd[0] = td.x; cd[1] = td.y; cd[2] = td.z; cd[3] = td.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = min(4 * q + e, C - 1);
          ca[e] = pa[c]; cb[e] = pb[c]; cq[e] = pc[c]; cd[e] = pd[c];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * q + e;
        if (c < C) {
          const bool off = c > on;
          const float a = off ? -1e10f : ca[e], b = off ? -1e10f : cb[e], cc = off ? -1e10f : cq[e], d = off ? -1e10f : cd[e];
          v[c] = wy0 * (wx0 * a + wx1 * b) + wy1 * (wx0 * cc + wx1 * d);        // logits_resize_kernel
          m = fmaxf(m, v[c]);
        }
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) { v[c] = expf(v[c] - m); s += v[c]; }                            // fuse_probs_kernel, A = 1
  int best = 0;
  float bv = -1.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) {
      // one augmentation: the mean over augmentations is (0 + pr) / 1.0f = pr exactly
      const float pr = v[c] / s;
      if (pr > bv) { bv = pr; best = c; }
    }
  return (float)best;
}

template <int MAXC>
__global__ void __launch_bounds__(256) frame_tail_kernel(const float* __restrict__ in, float* __restrict__ out4,
                                                         float* __restrict__ label_out, float* __restrict__ label_in, int IH, int IW,
                                                         int C, int ldi, int OH, int OW, int LH, int LW, int obj_total, int align,
                                                         float sh, float sw, int vec) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n_out = (long)OH * OW, n_in = label_in ? (long)LH * LW : 0, n4 = out4 ? (long)IH * IW * C : 0;
  const int on = group_obj_num(0, C, obj_total);
  if (idx < n_out) {
    const int oy = (int)(idx / OW), ox = (int)(idx - (long)oy * OW);
    label_out[idx] = tail_label_at<MAXC>(in, oy, ox, IH, IW, C, ldi, OH, OW, on, align, sh, sw, vec);
  } else if (idx < n_out + n_in) {
    const long i = idx - n_out;
    const int y = (int)(i / LW), x = (int)(i - (long)y * LW);
    // torch upsample_nearest2d ("nearest", legacy), as label_resize_kernel: src = min(floor(dst * (float)in / out), in - 1)
    const float rh = (float)OH / (float)LH, rw = (float)OW / (float)LW;
    const int sy = min((int)floorf(y * rh), OH - 1), sx = min((int)floorf(x * rw), OW - 1);
    label_in[i] = tail_label_at<MAXC>(in, sy, sx, IH, IW, C, ldi, OH, OW, on, align, sh, sw, vec);
  } else if (idx < n_out + n_in + n4) {
    const long i = idx - n_out - n_in;
    const long HW = (long)IH * IW;
    const int c = (int)(i / HW);
    const int p = (int)(i - (long)c * HW);
    out4[i] = (c > on) ? -1e10f : in[(long)p * ldi + c];                         // logits_planar_kernel
  }
}

extern "C" int aot_frame_tail_f32(const float* logits, float* out4, float* label_out, float* label_in, int IH, int IW, int C, int ldi,
                                  int OH, int OW, int LH, int LW, int obj_total, int align_corners, void* stream) {
  if (!logits || !label_out || IH <= 0 || IW <= 0 || C <= 1 || ldi < C || OH <= 0 || OW <= 0) return AOT_ERR_BADARG;
  if (label_in && (LH <= 0 || LW <= 0)) return AOT_ERR_BADARG;
  if (C > 16) return AOT_ERR_UNSUPPORTED;
  const int vec = (int)((ldi & 3) == 0 && ((uintptr_t)logits & 15) == 0 && ldi >= ((C + 3) & ~3));
  const long total = (long)OH * OW + (label_in ? (long)LH * LW : 0) + (out4 ? (long)IH * IW * C : 0);
  hipLaunchKernelGGL(frame_tail_kernel<16>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, logits, out4, label_out, label_in,
                     IH, IW, C, ldi, OH, OW, LH, LW, obj_total, align_corners, bilinear_scale(IH, OH, align_corners),
                     bilinear_scale(IW, OW, align_corners), vec);
  AOT_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Identity bank: one workgroup per output token; the KxK label patch is staged in LDS, then every
// thread (= channel quad) walks the taps, gathering coalesced rows of the [label, ky, kx, C] table.
// ---------------------------------------------------------------------------------------------
struct IdFuse {          // optional fused outputs: fout[i][row] = id_emb[row] + fadd[i][row]  (V + id_emb of layer i, transformer.py:366)
  const float* fadd[4];
  float* fout[4];
  int n, ldadd, ldout;
};

__global__ void __launch_bounds__(256) idbank_kernel(const float* __restrict__ mask, const float* __restrict__ table,
                                                     const float* __restrict__ sumtab, const float* __restrict__ bias,
                                                     float* __restrict__ out, int H, int W, int OW, int K, int stride,
                                                     int pad, int C, int nlabel, int ldo, int ntok, int group_size,
                                                     int group0, const IdFuse fz) {
  extern __shared__ int labels[];  // K*K entries (-1 = contributes nothing), then 3 x 64 float4 partial sums
  const int tok = blockIdx.x;
  // lane g of the batch = object group group0+g of the SAME label map (AOTInferEngine.separate_mask, aot_engine.py:515-534):
  // labels (group0+g)*group_size+1 .. +group_size become 1 .. group_size, every other pixel is background 0
  const int grp = blockIdx.y, lab_lo = (group0 + grp) * group_size;
  const long row = (long)grp * ntok + tok;
  const int Y = tok / OW, X = tok - Y * OW;
  const int KK = K * K;
  bool same = true;
  int first = -2;
  for (int i = threadIdx.x; i < KK; i += blockDim.x) {
    const int ky = i / K, kx = i - ky * K;
    const int iy = Y * stride - pad + ky, ix = X * stride - pad + kx;
    int lab = -1;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
      const float v = mask[(long)iy * W + ix];
      int li = (int)v;
      const bool integral = v == (float)li;
      if (group_size > 0 && integral) li = (li > lab_lo && li <= lab_lo + group_size) ? li - lab_lo : 0;
      if (integral && li >= 0 && li < nlabel) lab = li;
    }
    labels[i] = lab;
    if (first == -2) first = lab;
    same = same && (lab == first);
  }
  const int l0raw = __syncthreads_and(same ? 1 : 0);   // barrier + "every thread saw one label"
  const int l0 = labels[0];
  // every thread's subset is uniform; the window is uniform iff all subsets agree with labels[0]
  const bool uniform = sumtab != nullptr && l0 >= 0 && __syncthreads_and((first == -2 || first == l0) ? 1 : 0) && l0raw;
  // Fast path: the whole window carries ONE valid label (interiors of objects / background, i.e. most tokens of a
  // real mask): the K*K-tap sum is the precomputed per-label table sum.  Otherwise the 4 waves split the taps.
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  float4* psum = reinterpret_cast<float4*>(labels + ((KK + 3) & ~3));
  for (int c4 = lane; c4 < (C >> 2); c4 += 64) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (uniform) {
      if (part == 0) acc = *reinterpret_cast<const float4*>(sumtab + (long)l0 * C + c4 * 4);
    } else {
      for (int i = part; i < KK; i += 4) {
        const int lab = labels[i];
        if (lab < 0) continue;
        const float4 t = *reinterpret_cast<const float4*>(table + ((long)lab * KK + i) * C + c4 * 4);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
      if (part > 0) psum[(part - 1) * 64 + lane] = acc;
      __syncthreads();
      if (part == 0)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const float4 t = psum[q * 64 + lane];
          acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
      __syncthreads();
    }
    if (part == 0) {
      if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + c4 * 4);
        acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
      }
      if (out) *reinterpret_cast<float4*>(out + row * ldo + c4 * 4) = acc;
      for (int i = 0; i < fz.n; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(fz.fadd[i] + row * fz.ldadd + c4 * 4);
        *reinterpret_cast<float4*>(fz.fout[i] + row * fz.ldout + c4 * 4) = make_float4(acc.x + a.x, acc.y + a.y, acc.z + a.z, acc.w + a.w);
      }
    }
  }
}

extern "C" int aot_idbank_f32(const float* mask, const float* table, const float* sumtab, const float* bias, float* out,
                              int G, int group_size, int group0, int H, int W, int OH, int OW, int K, int stride, int pad,
                              int C, int nlabel, int ldo, const float* const* fuse_add, float* const* fuse_out, int nfuse,
                              int ldadd, int ldfout, void* stream) {
  if (!mask || !table || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || K <= 0 || C <= 0 || (C & 3) || (ldo & 3))
    return AOT_ERR_BADARG;
  if (G <= 0 || G > 65535 || group_size < 0 || group0 < 0 || ((G > 1 || group0 > 0) && group_size == 0)) return AOT_ERR_BADARG;
  if (nfuse < 0 || nfuse > 4 || (!out && nfuse == 0)) return AOT_ERR_BADARG;
  IdFuse fz;
  fz.n = nfuse; fz.ldadd = ldadd; fz.ldout = ldfout;
  for (int i = 0; i < 4; ++i) { fz.fadd[i] = nullptr; fz.fout[i] = nullptr; }
  if (nfuse > 0) {
    if (!fuse_add || !fuse_out || (ldadd & 3) || (ldfout & 3) || ldadd < C || ldfout < C) return AOT_ERR_BADARG;
    for (int i = 0; i < nfuse; ++i) {
      if (!fuse_add[i] || !fuse_out[i]) return AOT_ERR_BADARG;
      fz.fadd[i] = fuse_add[i];
      fz.fout[i] = fuse_out[i];
    }
  }
  hipLaunchKernelGGL(idbank_kernel, dim3(OH * OW, G), dim3(256), ((K * K + 3) & ~3) * sizeof(int) + 3 * 64 * sizeof(float4),
                     (hipStream_t)stream, mask, table, sumtab, bias, out, H, W, OW, K, stride, pad, C, nlabel, ldo, OH * OW,
                     group_size, group0, fz);
  AOT_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
  reinterpret_cast<float4*>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

extern "C" int aot_add_f32(const float* a, const float* b, float* out, long n, void* stream) {
  if (!a || !b || !out || n <= 0 || (n & 3)) return AOT_ERR_BADARG;
  hipLaunchKernelGGL(add_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4);
  AOT_LAUNCH_CHECK();
}

// Row-block copy between token-major buffers with a DEVICE-side destination slot (bank append inside a replayed graph):
// one thread per float4.
__global__ void __launch_bounds__(256) copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, long rows,
                                                        int C4, long src_brows, long dst_brows, int lds, int ldd,
                                                        const int* __restrict__ slot_dev, int slot) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long per = rows * C4;
  if (idx >= per * B) return;
  const int b = (int)(idx / per);
  const long rem = idx - (long)b * per;
  const long r = rem / C4;
  const int c = (int)(rem - r * C4) * 4;
  const long row0 = (long)(slot_dev ? *slot_dev : slot) * rows;
  const float4 v = *reinterpret_cast<const float4*>(src + ((long)b * src_brows + r) * lds + c);
  *reinterpret_cast<float4*>(dst + ((long)b * dst_brows + row0 + r) * ldd + c) = v;
}

extern "C" int aot_copy_rows_f32(const float* src, float* dst, int B, long rows, int C, long src_brows, long dst_brows, int lds,
                                 int ldd, const int* slot_dev, int slot, void* stream) {
  if (!src || !dst || B <= 0 || rows <= 0 || C <= 0 || (C & 3) || (lds & 3) || (ldd & 3) || slot < 0 || src_brows < 0 ||
      dst_brows < 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
    return AOT_ERR_BADARG;
  const long n = (long)B * rows * (C / 4);
  hipLaunchKernelGGL(copy_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, B, rows, C / 4, src_brows,
                     dst_brows, lds, ldd, slot_dev, slot);
  AOT_LAUNCH_CHECK();
}

extern "C" const char* aot_hip_version(void) { return "aot_hip 0.3 gfx950"; }
