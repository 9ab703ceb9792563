// What does the fp32 matrix pipe sustain when the same waves also feed it from LDS / run VALU work next to it?
// All variants: 4 accumulator chains per wave, 256 threads per workgroup, no global traffic in the loop.
//   MODE 0  operands in registers (the ceiling: mfma_data.hip)
//   MODE 1  operands read from LDS at the Conv1D-as-GEMM's rate: 3 ds_read_b128 per 8 MFMAs, every MFMA uses fresh data
//   MODE 2  operands in registers + attention-like VALU work: per 4 MFMAs one v_exp_f32 and 4 simple VALU ops
//   MODE 3  MODE 1 + MODE 2
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_mix.hip -o mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_mix(const float* in, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 a = *reinterpret_cast<const f32x4*>(in + lane * 4), b0 = *reinterpret_cast<const f32x4*>(in + 256 + lane * 4), b1 = *reinterpret_cast<const f32x4*>(in + 512 + lane * 4);
  f32x16 c[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
  float x = in[lane], acc = 0.f;
  const float* p = lds + lane * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {  // 8 MFMAs per slice
      if (MODE & 1) {
        a = *reinterpret_cast<const f32x4*>(p + ((it + s) & 7) * 256);
        b0 = *reinterpret_cast<const f32x4*>(p + 2048 + ((it + s) & 7) * 256);
        b1 = *reinterpret_cast<const f32x4*>(p + 4096 + ((it + s) & 7) * 256);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b0[e], c[0], 0, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b1[e], c[1], 0, 0, 0);
        if ((MODE & 2) && (e & 1)) {
          x = __builtin_amdgcn_exp2f(x * 0.5f - 1.0f);
          acc = fmaxf(acc, x) + x;
          x = x * 1.5f + acc * 0.25f;
          asm volatile("" : "+v"(x), "+v"(acc));
        }
      }
      // swap chains so all four accumulators are used
      f32x16 t = c[0]; c[0] = c[2]; c[2] = t; t = c[1]; c[1] = c[3]; c[3] = t;
    }
  }
  float s = acc + x;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const float* in, float* out, int blocks) {
  const int iters = 2000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_mix<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_mix<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  double fl = (double)blocks * 4 * iters * 64 * 4096.0;
  const char* what[] = {"registers only", "LDS-fed (3 b128 reads / 8 MFMAs)", "registers + exp/VALU", "LDS-fed + exp/VALU"};
  printf("%-36s %4d blocks (%d waves/SIMD): %7.2f ms  %.1f TF/s\n", what[MODE], blocks, blocks / 256, ms, fl / ms / 1e9);
}
int main() {
  float *in, *out;
  CK(hipMalloc(&in, 8192 * 4)); CK(hipMalloc(&out, 2048 * 256 * 4));
  static float h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = ((float)rand() / (float)RAND_MAX * 2 - 1);
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  for (int blocks : {256, 512}) { run<0>(in, out, blocks); run<1>(in, out, blocks); run<2>(in, out, blocks); run<3>(in, out, blocks); }
  return 0;
}
