// Kernel lab: fixed cost per k_attention workgroup.  256 workgroups (B16, H2, S=1024: one per CU), key length lens[b] = 32 k
// for k = 1..32, so the only thing that changes is the number of key tiles swept; time(k) = F + k * t_tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I../../smart-nar_fast_tts_amd/csrc attn_fixed.hip -o attn_fixed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "attention.hip"
namespace ns { bool launch_planner_enabled() { return true; } }  // (defined in gemm_conv.hip, which this harness does not link)
using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main() {
  const int B = 16, S = 1024, H = 2, dk = 128, d = H * dk;
  size_t n = (size_t)B * S * 3 * d;
  std::vector<float> h(n); for (auto& v : h) v = ((float)rand() / (float)RAND_MAX * 2 - 1);
  float *q, *o; long long* lens;
  CK(hipMalloc(&q, n * 4)); CK(hipMalloc(&o, n / 3 * 4)); CK(hipMalloc(&lens, B * 8)); CK(hipMemcpy(q, h.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int k : {1, 2, 4, 8, 16, 24, 32}) {
    std::vector<long long> l(B, 32ll * k); CK(hipMemcpy(lens, l.data(), B * 8, hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) CK(launch_attention(q, lens, B, S, H, dk, o, nullptr, 0, nullptr, 0));
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(a, 0)); for (int i = 0; i < 20; ++i) CK(launch_attention(q, lens, B, S, H, dk, o, nullptr, 0, nullptr, 0));
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms / 20 < best ? ms / 20 : best;
    }
    printf("key tiles %2d   %7.1f us   (MFMA time at 2.4 GHz: %5.1f us)\n", k, best * 1e3, k * 128 * 64 / 2400.0);
  }
  return 0;
}
