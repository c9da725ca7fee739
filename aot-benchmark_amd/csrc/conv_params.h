// Parameters shared by the implicit-GEMM convolution kernels (gemm_conv.hip: register-staged general kernel,
// gemm_lds.hip: LDS-direct tile kernel).
#pragma once
#include "common.h"

struct ConvParams {
  const float* in;    // NHWC activations of B images: [B*H*W, lda]
  const float* w;     // [K, ldb]   (k-major; general kernel)
  const float* wt;    // [Cout, ldwt] (k-contiguous rows; LDS-direct kernel) or nullptr
  const float* bias;  // [Cout] or nullptr
  const float* res;   // residual [res_rows, ldr] or nullptr; row = m % res_rows (a map shared by several lanes)
  float* out;         // [B*OH*OW, ldc]
  int B, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, dil;
  int lda, ldb, ldwt, ldc, ldr, res_rows, M, K, act;
};

// LDS-direct kernel (gemm_lds.hip).  variant: 0 = 128x64 tile, 1 = 64x64 tile (two DMA steps ahead); 2 / 3 = 64x64 with
// three / four steps ahead, 4 / 5 = 128x64 with three / four (for shapes with about one workgroup per CU).  ksplit > 1 cuts K into ksplit slices
// whose fp32 partial tiles go to `scratch` ([ksplit][M][Cout] floats) and are summed in slice order by a second launch.
int launch_gemm_lds(const ConvParams& p, int variant, int ksplit, float* scratch, hipStream_t s);
bool gemm_lds_eligible(const ConvParams& p);
// variant 6 / 7 = the lean 64x64 / 128x64 kernel (buffer-DMA addressing, interleaved issue; see gemm_lds.hip)
bool gemm_lean_eligible(const ConvParams& p);
// bf16 x 6 variant of the lean 64x64 kernel (gemm_lds.hip): w6 = the weight pre-split into three bf16 planes, layout
// [3][K/32][4][cout_pad][8] (aot_pack_bf16x6_f32)
bool gemm_x6_eligible(const ConvParams& p);
// tile: 0 = chosen by shape, 66 / 129 = forced (the 64x64 direct-weight form / the register-staged 128x128 form); terms 1 = plain bf16
int launch_gemm_x6(const ConvParams& p, const void* w6, int cout_pad, int tile, hipStream_t s, int terms = 6, int ksplit = 1, float* scratch = nullptr);
// the phase-shifted 128x128 form (gemm_x6pp_kernel) with split-K over the grid (ksplit >= 2, slabs in scratch)
// ln != nullptr (both split-K forms, Cout == 256): the reduce launch also writes LayerNorm(result) * gamma + beta to ln->out [M, ld]
struct LnOutArgs {
  const float* gamma;
  const float* beta;
  float* out;
  int ld;
  float eps;
};
int launch_gemm_x6pp(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s, int ksplit, float* scratch, const LnOutArgs* ln = nullptr);
// split-K over the grid on the 64x64 register-staged kernel with the weight fragments straight from global memory
int launch_gemm_x6rd_splitk(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s, int ksplit, float* scratch, const LnOutArgs* ln = nullptr);
// a linear layer on the same kernel whose tile end also writes GroupNorm partial sums: gn_part [2 * ceil(M / 64)][Cout / 32][2] floats
int launch_gemm_x6rd_gn(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s, float* gn_part);
// LayerNorm over the K channels of `in` + the linear layer in one launch (gamma / beta folded into w6 / bias by the caller); gn_part
// optional (the GroupNorm partials of the output as launch_gemm_x6rd_gn writes them)
int launch_gemm_x6rd_ln(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s, float eps, const float* colsum, float* gn_part);
// n <= 4 linear layers of ONE shape (p: M, K, Cout, leading dimensions, res_rows, act) in one launch; per-problem operand pointers
int launch_gemm_x6rd_group(const ConvParams& p, int n, const float* const* in, const void* const* w6, const float* const* bias,
                           const float* const* res, float* const* out, int cout_pad, hipStream_t s);
// a KxK convolution on four input channels (Cin = lda = 4: the ResNet stem) on the same kernel; w6: K rounded up to 8 taps per k-step
int launch_gemm_x6rd_c4(const ConvParams& p, const void* w6, int cout_pad, hipStream_t s);
