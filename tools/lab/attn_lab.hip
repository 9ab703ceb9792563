// Kernel lab: times k_attention on the path's shapes, standalone (no torch).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I../../smart-nar_fast_tts_amd/csrc attn_lab.hip -o attn_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "attention.hip"
namespace ns { bool launch_planner_enabled() { return true; } }  // (defined in gemm_conv.hip, which this harness does not link)
using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main() {
  struct Shape { const char* name; int B, S, H, dk; } shapes[] = {
    {"cfg2 dec  B16 S1010 H2 dk128", 16, 1010, 2, 128}, {"cfg5 dec  B8 S3880 H2 dk128", 8, 3880, 2, 128},
    {"cfg4 dec  B64 S1045 H8 dk64 ", 64, 1045, 8, 64},  {"cfg1 dec  B1 S788 H2 dk128 ", 1, 788, 2, 128},
    {"cfg2 enc  B16 S128 H2 dk128 ", 16, 128, 2, 128}};
  for (auto& s : shapes) {
    const int d = s.H * s.dk; size_t n = (size_t)s.B * s.S * 3 * d;
    std::vector<float> h(n); for (auto& v : h) v = ((float)rand() / RAND_MAX * 2 - 1);
    float *q, *o; CK(hipMalloc(&q, n * 4)); CK(hipMalloc(&o, n / 3 * 4)); CK(hipMemcpy(q, h.data(), n * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) CK(launch_attention(q, nullptr, s.B, s.S, s.H, s.dk, o, nullptr, 0, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0)); for (int i = 0; i < 10; ++i) CK(launch_attention(q, nullptr, s.B, s.S, s.H, s.dk, o, nullptr, 0, nullptr, 0));
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    double gf = 4.0 * s.B * s.H * (double)s.S * s.S * s.dk / 1e9;
    printf("%s  %7.2f GFLOP  %8.1f us  %6.1f TF/s\n", s.name, gf, ms * 1e3, gf / ms);
    CK(hipFree(q)); CK(hipFree(o));
  }
  return 0;
}
