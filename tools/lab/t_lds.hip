#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* x, float* y, int n, int soff) {
  __shared__ __attribute__((aligned(16))) float buf[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) buf[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, (short)0, 0x7FFFFFFF, 0x00020000);
  int voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0x80000000;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)buf, 16, voff, soff, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) y[i] = buf[i];
}
int main() {
  float *x, *y; hipMalloc(&x, 4096); hipMalloc(&y, 1024);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1; hipMemcpy(x, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, 1024, 16);
  float o[256]; hipMemcpy(o, y, 1024, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) printf("%g ", o[i]); printf("\n");
  return 0;
}
