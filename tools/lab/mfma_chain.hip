// Does ONE wave per SIMD keep the fp32 matrix pipe full with a single dependent accumulator chain?
// (k_attention's score tile is one chain of DK/2 MFMAs; the GEMMs interleave >= 2 accumulators and >= 2 waves.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NC>
__global__ __launch_bounds__(256) void k_chain(const float* in, float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) % 4096]; b[i] = in[(threadIdx.x * 8 + i + 17) % 4096]; }
  f32x16 c[NC];
  for (int k = 0; k < NC; ++k) for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 32; ++e) c[e % NC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e & 7], b[(e + e / 8) & 7], c[e % NC], 0, 0, 0);
  }
  float s = 0;
  for (int k = 0; k < NC; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NC> void run(const float* in, float* out, int blocks) {
  const int iters = 4000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_chain<NC>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_chain<NC>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  double fl = (double)blocks * 4 * iters * 32 * 4096.0;
  printf("%d accumulator chain(s), %4d blocks (%d waves/SIMD): %.2f ms  %.1f TF/s\n", NC, blocks, blocks / 256, ms, fl / ms / 1e9);
}
int main() {
  float *in, *out;
  CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 2048 * 256 * 4));
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = ((float)rand() / RAND_MAX * 2 - 1) * 0.01f;
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  for (int blocks : {256, 512}) { run<1>(in, out, blocks); run<2>(in, out, blocks); run<4>(in, out, blocks); }
  return 0;
}
