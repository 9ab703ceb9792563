// Is the fp32 MFMA ceiling data dependent?  Same instruction stream (4 accumulator chains, operands in registers), three
// operand sets: zeros, 8 distinct random values reused, 64 distinct random values per lane cycled through.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_data.hip -o mfma_data
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NV>
__global__ __launch_bounds__(256) void k_data(const float* in, float* out, int iters) {
  float a[NV], b[NV];
  for (int i = 0; i < NV; ++i) { a[i] = in[(threadIdx.x * NV + i) % 65536]; b[i] = in[(threadIdx.x * NV + i + 17 * NV) % 65536]; }
  f32x16 c[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 64; ++e) c[e & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e % NV], b[(e * 7 + 3) % NV], c[e & 3], 0, 0, 0);
  }
  float s = 0;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV> void run(const float* in, float* out, int blocks, const char* what) {
  const int iters = 4000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_data<NV>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_data<NV>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  double fl = (double)blocks * 4 * iters * 64 * 4096.0;
  printf("%-40s %2d values/lane, %4d blocks (%d waves/SIMD): %7.2f ms  %.1f TF/s\n", what, NV, blocks, blocks / 256, ms, fl / ms / 1e9);
}
int main() {
  float *in, *out;
  CK(hipMalloc(&in, 65536 * 4)); CK(hipMalloc(&out, 2048 * 256 * 4));
  static float h[65536];
  for (int mode = 0; mode < 3; ++mode) {
    for (int i = 0; i < 65536; ++i) h[i] = mode == 0 ? 0.f : mode == 1 ? ((float)rand() / (float)RAND_MAX * 2 - 1) * 0.01f : ((float)rand() / (float)RAND_MAX * 2 - 1);
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    const char* what = mode == 0 ? "zeros" : mode == 1 ? "random, |x| < 0.01" : "random, |x| < 1";
    for (int blocks : {256, 512}) { run<8>(in, out, blocks, what); run<64>(in, out, blocks, what); }
  }
  return 0;
}
