// fp32 MFMA rate of the 16-row shape, v_mfma_f32_16x16x4_f32 (4 accumulator registers, 16x16 output, k = 4 per instruction),
// beside the path's v_mfma_f32_32x32x2_f32 (16 accumulator registers, 32x32, k = 2): the shape a 16-row quantum for the
// full-row / remainder tiles would need (DESIGN.md "what comes next").  Same loop skeleton as mfma_peak.hip.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_16x16 mfma_16x16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int SHAPE>  // 32: 32x32x2, 16: 16x16x4
__global__ __launch_bounds__(256) void k_peak(const float* in, float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) % 4096]; b[i] = in[(threadIdx.x * 8 + i + 17) % 4096]; }
  float s = 0;
  if (SHAPE == 32) {
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
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  } else {
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[(e + 1) & 7], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(e + 1) & 7], b[e], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(e + 2) & 7], b[(e + 3) & 7], c3, 0, 0, 0);
      }
    }
    for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float *in, *out;
  CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 2048 * 256 * 4));
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = ((float)rand() / RAND_MAX * 2 - 1) * 0.01f;
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  for (int shape : {32, 16}) {
    for (int blocks : {256, 512, 1024}) {
      const int iters = 4000;
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      auto go = [&] { if (shape == 32) hipLaunchKernelGGL(k_peak<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters); else hipLaunchKernelGGL(k_peak<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters); };
      go();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a, 0));
      for (int r = 0; r < 5; ++r) go();
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
      // flops per instruction: 32x32x2 -> 2*32*32*2 = 4096; 16x16x4 -> 2*16*16*4 = 2048
      const double fl = (double)blocks * 4 * iters * 32 * (shape == 32 ? 4096.0 : 2048.0);
      printf("v_mfma_f32_%s, %4d blocks (%d waves/SIMD): %.2f ms  %.1f TF/s\n", shape == 32 ? "32x32x2" : "16x16x4", blocks, blocks / 256, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
