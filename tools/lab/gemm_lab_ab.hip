// Kernel lab, A/B harness: times launch_conv_gemm's own choice on the path's big GEMM shapes.  Build it twice with
// different -D switches of gemm_conv.hip (or two checkouts) and run both binaries in ONE gpurun call: only same-run
// numbers compare.   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DNS_...] -I../../smart-nar_fast_tts_amd/csrc gemm_lab_ab.hip -o gemm_lab_ab_X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_conv.hip"

using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3;
  struct Shape { const char* name; int M, S, Cin, KW, N; } shapes[] = {
    {"ffn_w1 dec  (k9 256->1024) M16160", 16160, 1010, 256, 9, 1024},
    {"postnet mid (k5 512->512)  M16160", 16160, 1010, 512, 5, 512},
    {"ffn_w2 dec  (k1 1024->256) M16160", 16160, 1010, 1024, 1, 256},
    {"fc dec      (k1 256->256)  M16160", 16160, 1010, 256, 1, 256},
    {"qkv dec     (k1 256->768)  M16160", 16160, 1010, 256, 1, 768},
    {"pred conv   (k3 256->256)  M16160", 16160, 1010, 256, 3, 256},
    {"ffn_w1 enc  (k9 256->1024) M2048 ", 2048, 128, 256, 9, 1024},
    {"ffn_w1 d512 (k9 512->1024) M64640", 64640, 1010, 512, 9, 1024},
    {"qkv d512    (k1 512->1536) M64640", 64640, 1010, 512, 1, 1536},
    {"fc d512     (k1 512->512)  M64640", 64640, 1010, 512, 1, 512},
    {"ffn_w2 d512 (k1 1024->512) M64640", 64640, 1010, 1024, 1, 512},
    {"postnet mid (k5 512->512)  M64640", 64640, 1010, 512, 5, 512},
    {"postnet L   (k5 512->80)   M16160", 16160, 1010, 512, 5, 80},
    {"postnet L   (k5 512->80)   M31200", 31200, 3900, 512, 5, 80},
    {"postnet L   (k5 512->80)   M64640", 64640, 1010, 512, 5, 80},
    {"mel_linear  (k1 256->80)   M16160", 16160, 1010, 256, 1, 80},
    {"ffn_w1 long (k9 256->1024) M31200", 31200, 3900, 256, 9, 1024},
    {"postnet mid (k5 512->512)  M31200", 31200, 3900, 512, 5, 512},
  };
  for (auto& s : shapes) {
    size_t nx = (size_t)s.M * s.Cin, nw = (size_t)s.N * s.KW * s.Cin, ny = (size_t)s.M * s.N;
    std::vector<float> hx(nx), hw(nw), hb(s.N);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    for (auto& v : hb) v = (float)rand() / RAND_MAX;
    float *dx, *dw, *db, *dy;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dy, ny * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice));
    ConvGemm p; memset(&p, 0, sizeof(p)); p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.resid = nullptr; p.ldr = 0; p.Y = dy; p.ldy = s.N;
    p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = ACT_RELU;
    double gf = 2.0 * s.M * s.Cin * s.KW * s.N / 1e9;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) CK(launch_conv_gemm(p, 0));
    CK(hipDeviceSynchronize());
    printf("%s %6.1f GFLOP:", s.name, gf);
    for (int r = 0; r < rounds; ++r) {
      const int iters = 20;
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < iters; ++i) CK(launch_conv_gemm(p, 0));
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= iters;
      printf("  %7.1f us %6.1f TF/s", ms * 1e3, gf / ms);
    }
    printf("\n");
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
  }
  return 0;
}
