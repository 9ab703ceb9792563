// fp32 MFMA ceiling under load: pure v_mfma_f32_32x32x2_f32 loop on random (or zero) operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_peak(const float* in, float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) % 4096]; b[i] = in[(threadIdx.x * 8 + i + 17) % 4096]; }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[(e + 1) & 7], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(e + 1) & 7], b[e], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(e + 2) & 7], b[(e + 3) & 7], c3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float *in, *out;
  CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 2048 * 256 * 4));
  for (int mode = 0; mode < 2; ++mode) {
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = mode ? ((float)rand() / RAND_MAX * 2 - 1) * 0.01f : 0.f;
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    for (int blocks : {256, 512, 1024}) {
      const int iters = 4000;
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      hipLaunchKernelGGL(k_peak, dim3(blocks), dim3(256), 0, 0, in, out, iters);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a, 0));
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_peak, dim3(blocks), dim3(256), 0, 0, in, out, iters);
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
      double fl = (double)blocks * 4 * iters * 32 * 4096.0;
      printf("%s data, %4d blocks (%d waves/SIMD): %.2f ms  %.1f TF/s\n", mode ? "random" : "zero  ", blocks, blocks / 256, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
