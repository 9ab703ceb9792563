// Kernel lab (round 4): k_attention + k_attention_merge with a FORCED key split on dense launches that fill the chip
// unevenly (B = 9 ... 24 utterances of ~1010 frames): is a launch's time rounds-of-256 x sweep length, and what does the
// merge cost?   Input for attention_split() in attention.hip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I../../smart-nar_fast_tts_amd/csrc attn_lab_split.hip -o attn_lab_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "attention.hip"
namespace ns { bool launch_planner_enabled() { return true; } }  // (defined in gemm_conv.hip, which this harness does not link)
using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int DK>
static void run(const float* q, int B, int S, int H, float* o, float* scratch, int nsplit, hipStream_t st) {
  const int d = H * DK, qtiles = (S + 127) / 128;
  const size_t M = (size_t)B * S;
  const float c = 1.4426950408889634f / sqrtf((float)DK);
  float* opart = nsplit > 1 ? scratch : nullptr;
  float* mlpart = nsplit > 1 ? scratch + (size_t)nsplit * M * d : nullptr;
  hipLaunchKernelGGL((k_attention<DK>), dim3(qtiles * nsplit, H, B), dim3(256), 0, st, q, nullptr, S, d, c, o, nsplit, opart, mlpart, nullptr, nullptr, nullptr, nullptr, 0, 0, 0);
  if (nsplit > 1) hipLaunchKernelGGL(k_attention_merge, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, opart, mlpart, (int)M, d, H, DK, nsplit, o);
}

int main() {
  struct Shape { int B, S, H, dk; } shapes[] = {{5, 1010, 2, 128}, {6, 1010, 2, 128}, {9, 1010, 2, 128}, {10, 1010, 2, 128}, {12, 1010, 2, 128}, {14, 1010, 2, 128}, {16, 1010, 2, 128},
                                               {17, 1010, 2, 128}, {18, 1010, 2, 128}, {20, 1010, 2, 128}, {24, 1010, 2, 128}, {28, 1010, 2, 128}, {32, 1015, 2, 128},
                                               {5, 3900, 2, 128}, {10, 3900, 2, 128}, {20, 1041, 8, 64}, {40, 1041, 8, 64}};
  for (auto& s : shapes) {
    const int d = s.H * s.dk; size_t n = (size_t)s.B * s.S * 3 * d; const size_t M = (size_t)s.B * s.S;
    std::vector<float> h(n); for (auto& v : h) v = ((float)rand() / RAND_MAX * 2 - 1);
    float *q, *o, *sc; CK(hipMalloc(&q, n * 4)); CK(hipMalloc(&o, n / 3 * 4)); CK(hipMalloc(&sc, (size_t)16 * (M * d + 2 * M * s.H) * 4));
    CK(hipMemcpy(q, h.data(), n * 4, hipMemcpyHostToDevice));
    const int tiles = (s.S + 31) / 32, blocks = ((s.S + 127) / 128) * s.H * s.B;
    printf("B=%d S=%d H=%d dk=%d  blocks=%d tiles=%d:", s.B, s.S, s.H, s.dk, blocks, tiles);
    for (int ns : {1, 2, 3, 4, 5, 6, 8, 11, 16}) {
      const int tps = (tiles + ns - 1) / ns;
      if ((tiles + tps - 1) / tps != ns) continue;  // no empty ranges
      auto go = [&] { if (s.dk == 128) run<128>(q, s.B, s.S, s.H, o, sc, ns, 0); else run<64>(q, s.B, s.S, s.H, o, sc, ns, 0); };
      for (int i = 0; i < 2; ++i) go();
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a, 0)); for (int i = 0; i < 10; ++i) go();
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
      printf("  n%d(%dwg,%dt):%.1f", ns, blocks * ns, tps, ms * 1e3);
    }
    printf("\n");
    CK(hipFree(q)); CK(hipFree(o)); CK(hipFree(sc));
  }
  return 0;
}
