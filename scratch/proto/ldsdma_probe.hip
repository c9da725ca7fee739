// Where does global_load_lds_dwordx4 put each lane's 16 bytes?  Prints, for the first lanes, the LDS float index at which
// the global float index landed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const float* __restrict__ g, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float lds[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = -1.f;
  __syncthreads();
  __builtin_amdgcn_global_load_lds((gptr_t)(g + lane * 4), (lptr_t)lds, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0xF70);
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  float h[256], o[1024]; for (int i = 0; i < 256; ++i) h[i] = (float)i;
  float *d, *r; hipMalloc(&d, sizeof(h)); hipMalloc(&r, sizeof(o)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, r); hipDeviceSynchronize();
  hipMemcpy(o, r, sizeof(o), hipMemcpyDeviceToHost);
  printf("LDS[0..23]: "); for (int i = 0; i < 24; ++i) printf("%g ", o[i]); printf("\n");
  printf("LDS[60..70]: "); for (int i = 60; i < 70; ++i) printf("%g ", o[i]); printf("\n");
  printf("LDS[252..262]: "); for (int i = 252; i < 262; ++i) printf("%g ", o[i]); printf("\n");
  int contiguous = 1; for (int i = 0; i < 256; ++i) if (o[i] != (float)i) contiguous = 0;
  int planar = 1; for (int l = 0; l < 64; ++l) for (int c = 0; c < 4; ++c) if (o[c * 64 + l] != (float)(l * 4 + c)) planar = 0;
  printf("lane-contiguous (lane*16 B): %d   dword-planar (c*256 B + lane*4): %d\n", contiguous, planar);
  return 0;
}
